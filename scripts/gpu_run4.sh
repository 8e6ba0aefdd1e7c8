#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout 600 python scripts/conv_bench.py > gpurun_out/conv_bench.log 2>&1
cat gpurun_out/conv_bench.log
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc1 -o pmc --output-format csv -- python $R/scripts/conv_bench.py quick > $R/gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc2 -o pmc --output-format csv -- python $R/scripts/conv_bench.py quick > $R/gpurun_out/pmc2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ('pmc1', 'pmc2'):
    for f in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            if 'gemm' not in k: continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in acc.items():
            print(d, k, {n: f'{x:.3g}' for n, x in v.items()})
PY
tail -5 gpurun_out/pmc1.log
