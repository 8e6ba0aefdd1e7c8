#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout 600 python scripts/conv_tiles.py 2>&1 | grep -v amdgpu
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --kernel-trace -d $R/gpurun_out/pmc1 -o pmc --output-format csv -- python $R/scripts/conv_bench.py quick > $R/gpurun_out/pmc1.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc1/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'gemm_bf16' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in acc.items():
        print(k, {n: f'{x:.3g}' for n, x in v.items()})
        if v.get('SQ_INSTS_MFMA'):
            print('   per MFMA: VALU %.1f SALU %.1f; wave cycles: wait %.0f%% wait_inst %.0f%% active %.0f%%' % (
                v['SQ_INSTS_VALU'] / v['SQ_INSTS_MFMA'], v['SQ_INSTS_SALU'] / v['SQ_INSTS_MFMA'],
                100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], 100 * v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'], 100 * v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']))
PY
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
