#!/bin/bash
# PMC passes for the EM imputation kernel (1000^3, R = 20, 20 % and 1 % missing; separate runs per counter, see
# MI355X_MICROARCH.md): HBM bytes per launch of em_cp_vec_k<float,4,20,true> (imputation + fused partial contraction).
# usage: pmc_em.sh <round tag, e.g. r03>
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_em
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; frac=$2; shift; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/time_em.py $frac > $OUT/$name.log 2>&1; }
for frac in 0.2 0.01; do
run fetch_$frac $frac FETCH_SIZE
run write_$frac $frac WRITE_SIZE
done
python3 - <<PY
import csv, glob, json, os, collections
out = '$OUT'
def collect(sub, kname):
    f = sorted(glob.glob(os.path.join(out, sub, '*', '*_counter_collection.csv')), key=os.path.getmtime)[-1]
    acc = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(f)):
        if kname not in r['Kernel_Name']:
            continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    n = max(len(disp), 1)
    return {k: v / n for k, v in acc.items()}, n
res = {}
for frac in ('0.2', '0.01'):
    p = float(frac)
    lines = 1.0 - (1.0 - p) ** 32                      # share of 128-B lines (32 fp32 entries) with a missing entry
    algo = 4e9 + 0.125e9 + 4e9 * lines + 0.08e9
    fe, nf = collect('fetch_' + frac, 'em_cp_vec_k<float, 4, 20, true>')
    wr, nw = collect('write_' + frac, 'em_cp_vec_k<float, 4, 20, true>')
    raw = fe.get('FETCH_SIZE', 0.0); w = wr.get('WRITE_SIZE', 0.0)
    unit = 1024.0 if 0 < raw < 1e9 else 1.0
    res['missing_' + frac] = {'FETCH_SIZE_bytes_raw': raw * unit, 'FETCH_SIZE_bytes_corrected_x2': 2 * raw * unit, 'WRITE_SIZE_bytes': w * unit,
                              'hbm_traffic_bytes_per_launch': 2 * raw * unit + w * unit, 'algorithmic_bytes_per_launch': algo,
                              'lines_with_a_missing_entry': lines, 'launches_sampled': {'fetch': nf, 'write': nw}}
res['workload'] = '1000^3 fp32 tensor, R = 20 (tools/time_em.py), fused update pass em_cp_vec_k<float,4,20,true>; algorithmic bytes: tensor 4 GB + mask 0.125 GB (one bit per entry) read, the 128-B lines with a missing entry written, T 0.08 GB written'
res['note'] = 'gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md HBM section): doubled'
json.dump(res, open('$R/gpurun_out/${TAG}_pmc_em_cp_vec.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/*/
