cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do for v in main nfwprobe; do
  if [ $v = main ]; then unset HMG_LIB_PATH; else export HMG_LIB_PATH=$PWD/hmvec_amd/libhmgrid_$v.so; fi
  python3 bench.py --no-cpu-baseline --no-limber --no-readme --no-long-grid --steps 40 > /tmp/ab_$v.json 2>/dev/null
  python3 - $v <<'PY'
import json, sys
d = json.loads(open(f"/tmp/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k = d["kernels"]
print(f"{sys.argv[1]:9s} step {d['ms_per_step']:.4f}  nfw {k['nfw_kernel']['ms']:.4f}")
PY
done; done
