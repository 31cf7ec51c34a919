#!/bin/bash
# the driver's command on the GPU box, one summary line: value, ms_per_step, roofline.frac, raygen / trace / tail ms, roofline.commit, kernel_commit, bytes of the headline line
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file /dev/null 2>/dev/null | tail -1 > /tmp/headline.json
python - <<'PY'
import json
l = open("/tmp/headline.json").read()
d = json.loads(l); r = d["roofline"]
print("%s: %s %s %s %s %s %s %s %s %d" % (d.get("kernel_commit"), d["value"], d["ms_per_step"], r["frac"], *[(r.get("kernel_ms") or {}).get(k) for k in ("raygen", "trace", "tail")], r.get("commit"), d.get("kernel_commit"), len(l.strip())))
PY
