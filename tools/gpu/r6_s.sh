#!/bin/bash
# Round 6, call S: counters and tables of the final build -- fetch / write counters of the volume builder and the march forward
# kernel STAMPED with the sha256 of their sources (bench.py's `traffic` constants; a stale stamp -> null), counters of the 2-D
# convolution kernel, the cold kernel table, the GPU suite once more in the driver's form (another box), the headline line.
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
L=gpurun_out/r6s
R=$PWD
rm -f ${L}_pmc_*.txt
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 5 --only cost_volume > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x cost_volume_fwd >> ${L}_pmc_cost_volume_fwd.txt 2>&1
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only conv_32_32_L0_fwd > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x marchw >> ${L}_pmc_conv3d_marchw.txt 2>&1
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 5 --only conv2d_64_64 > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x conv2d_march >> ${L}_pmc_conv2d.txt 2>&1
done
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 5 --only conv2d_64_64 > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x conv2d_march >> ${L}_pmc_conv2d.txt 2>&1
done
python -c "import bench; print('kernel_source_sha', bench.kernel_source_sha('cost_volume_mfma.hip'))" >> ${L}_pmc_cost_volume_fwd.txt
python -c "import bench; print('kernel_source_sha', bench.kernel_source_sha('conv3d.hip'))" >> ${L}_pmc_conv3d_marchw.txt
python -c "import bench; print('kernel_source_sha', bench.kernel_source_sha('conv2d.hip'))" >> ${L}_pmc_conv2d.txt
cp ${L}_pmc_cost_volume_fwd.txt profiles/r06_pmc_cost_volume_fwd.txt; cp ${L}_pmc_conv3d_marchw.txt profiles/r06_pmc_conv3d_marchw.txt; cp ${L}_pmc_conv2d.txt profiles/r06_pmc_conv2d.txt
cat ${L}_pmc_cost_volume_fwd.txt ${L}_pmc_conv3d_marchw.txt ${L}_pmc_conv2d.txt | cut -c1-120
timeout 400 python tools/kernel_bench.py --cold --iters 10 > ${L}_kernel_bench_cold.jsonl 2>/dev/null; wc -l ${L}_kernel_bench_cold.jsonl
rm -f gpurun_out/parity_report.jsonl
( time timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=15 2>&1 | grep -v "^  " | tail -60 ) > ${L}_pytest.log 2>&1; tail -6 ${L}_pytest.log | cut -c1-200
cp gpurun_out/parity_report.jsonl ${L}_parity_report.jsonl 2>/dev/null
timeout 700 python bench.py 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc_train.json; cut -c1-170 ${L}_bench_gwc_train.json
