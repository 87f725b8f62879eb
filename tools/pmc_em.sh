#!/bin/bash
# PMC passes for the EM imputation kernel (1000^3, R = 20, 20 % missing; separate runs per counter, see
# MI355X_MICROARCH.md): HBM bytes per launch of em_cp_vec_k<float,4,20,true> (imputation + fused partial contraction)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_em
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/time_em.py > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
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
# the two variants are separated by their template argument in the kernel name
for tag, key, algo in (('fused_update', 'true>', 4e9 + 1e9 + 4e9 + 0.08e9), ('unfused_mixed', 'false>', None)):
    fe, nf = collect('fetch', 'em_cp_vec_k<float, 4, 20, ' + key)
    wr, nw = collect('write', 'em_cp_vec_k<float, 4, 20, ' + key)
    raw = fe.get('FETCH_SIZE', 0.0); w = wr.get('WRITE_SIZE', 0.0)
    unit = 1024.0 if 0 < raw < 1e9 else 1.0
    res[tag] = {'FETCH_SIZE_bytes_raw': raw * unit, 'FETCH_SIZE_bytes_corrected_x2': 2 * raw * unit, 'WRITE_SIZE_bytes': w * unit,
                'hbm_traffic_bytes_per_launch': 2 * raw * unit + w * unit, 'algorithmic_bytes_per_launch': algo,
                'launches_sampled': {'fetch': nf, 'write': nw}}
res['unfused_mixed']['note'] = 'em_cp_vec_k<...,false>: the statistics-only passes (no write-back, 5 GB) and the un-fused update pass of each solve\'s last iteration (9 GB) share this kernel name; per-launch averages mix them'
res['workload'] = '1000^3 fp32 tensor, R = 20, 20 % missing (tools/time_em.py); algorithmic bytes: tensor 4 GB + mask 1 GB read, tensor 4 GB written (every vector), T 0.08 GB written'
res['note'] = 'gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md HBM section): doubled'
json.dump(res, open('$R/gpurun_out/r02_pmc_em_cp_vec.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/*/
