#!/bin/bash
# Round-5 evidence for k_wgrad_pc in ONE GPU call (variant builds tools/libexp_<TAG>.so must exist: tools/mkvar.sh):
#   gpurun_out/r05_wgrad_experiments.txt  A/B against k_wgrad_x6 and the timing-only / alternative builds
#   gpurun_out/r05_wgrad_timeline.txt     s_memtime timelines, single launch and the last of 100 back-to-back launches
#   gpurun_out/r05_wgrad_pmc.csv          SQ counters of the kernel (tools/pmc_wgrad.sh)
cd "$(dirname "$0")/.."
O=gpurun_out
{
  echo "# tools/kbench_wgrad.py: the four E-row weight-gradient jobs of a round (E = 180 082, 738 MB of operand rows) + k_wgrad_red,"
  echo "# min over 5 rounds of the mean of 20 back-to-back calls (HIP events); alternating, 2 rounds, one GPU call"
  for rep in 1 2; do
    MGN_WGRAD_PC=1 python tools/kbench_wgrad.py 16 | sed 's/shipped build/k_wgrad_pc (default)/'
    MGN_WGRAD_PC=0 python tools/kbench_wgrad.py 16 | sed 's/shipped build/k_wgrad_x6<6> (MGN_WGRAD_PC=0)/'
    for v in NOSPLIT NOMFMA NOLOAD NOSM VSPLIT M32 FLAGS; do
      [ -f tools/libexp_$v.so ] && MGN_LIB=tools/libexp_$v.so python tools/kbench_wgrad.py 16
    done
  done 2>&1 | grep "us"
  echo "# NOSPLIT / NOMFMA (two of six terms) / NOLOAD (cache-resident rows) / NOSM (neither splits nor terms): timing-only builds, results invalid by design"
  echo "# VSPLIT: the producers' split on the vector pipe; M32: v_mfma_f32_32x32x16_bf16 consumers; FLAGS: point-to-point flags instead of s_barrier"
} > $O/r05_wgrad_experiments.txt
{
  echo "# tools/timeline_wpc.py (build -DMGN_TIMELINE): s_memtime stamps of wave 0 (producer of A) and wave 4 (consumer) of workgroup 0; shader cycles"
  echo "# producer tags: 0 loop top, 3 next loads issued, 1 rows arrived + column sums, 2 split + piece writes issued; consumer: 0 after the barrier, 1 MFMAs + reads issued"
  echo "== single launches (synchronised)"; TL_REPS=3 python tools/timeline_wpc.py
  echo "== the last of 100 back-to-back launches"; TL_REPS=100 python tools/timeline_wpc.py
} 2>&1 | grep -v amdgpu.ids > $O/r05_wgrad_timeline.txt
tools/pmc_wgrad.sh > /dev/null 2>&1
cp $O/pmc_wgrad/wgrad_pmc.csv $O/r05_wgrad_pmc.csv
tail -5 $O/r05_wgrad_experiments.txt; tail -12 $O/r05_wgrad_timeline.txt; head -3 $O/r05_wgrad_pmc.csv | cut -c1-200
