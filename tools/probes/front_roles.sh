cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for nz in 32 4; do for v in a b c; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/fr_${v}_$nz -o k --output-format csv -- python3 tools/probes/front_roles.py $v $nz > gpurun_out/fr_${v}_$nz.log 2>&1
  python3 - gpurun_out/fr_${v}_$nz/k_kernel_stats.csv $v $nz <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'front_group' in r['Name'] or 'sigma2' in r['Name'] or 'halo' in r['Name']:
        print(sys.argv[2], 'nz=' + sys.argv[3], r['Calls'], round(float(r['AverageNs'])/1e3, 1), r['Name'][:40])
PY
done; done
