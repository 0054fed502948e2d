#!/bin/bash
# Round 5, call C: the mixed read/write stream ceiling (tools/ubench/store_stream), the whole GPU suite in the driver's form on the
# build with the f-1 kernels and the ACV cut behind concatconv, the headline bench line, fetch / write counters of the volume
# builder and of the march forward kernel (bench.py's `traffic` constants of this round).
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5c
R=$PWD
( timeout 120 tools/ubench/store_stream 2>&1 ) > ${L}_store_stream.txt; grep -E "mix|fill units nt    1|copy" ${L}_store_stream.txt
rm -f gpurun_out/parity_report.jsonl
( timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=8 2>&1 | grep -v "^  " | tail -70 ) > ${L}_pytest.log 2>&1; tail -14 ${L}_pytest.log | cut -c1-300
timeout 400 python bench.py --config gwc_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc.json; cut -c1-260 ${L}_bench_gwc.json
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 5 --only cost_volume > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x cost_volume_fwd >> ${L}_pmc_cost_volume_fwd.txt 2>&1
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only conv_32_32_L0_fwd > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x marchw >> ${L}_pmc_conv3d_marchw.txt 2>&1
done
cat ${L}_pmc_cost_volume_fwd.txt ${L}_pmc_conv3d_marchw.txt | cut -c1-120
