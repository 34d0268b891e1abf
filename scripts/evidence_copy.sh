#!/bin/bash
# Copies what scripts/evidence.sh left under gpurun_out/<tag>/ into profiles/ under this round's names.
# usage: scripts/evidence_copy.sh <tag> [round prefix, default r04]
T=gpurun_out/${1:-r04ev}
R=${2:-r04}
for c in cfg1 cfg2 cfg3 cfg4 cfg5; do
  [ -s $T/bench_$c.json ] && cp $T/bench_$c.json profiles/${R}_bench_${c}_n1.json
  [ -s $T/rocprofv3_kernel_stats_$c.csv ] && cp $T/rocprofv3_kernel_stats_$c.csv profiles/${R}_rocprofv3_kernel_stats_$c.csv
  [ -s $T/pmc_$c.json ] && cp $T/pmc_$c.json profiles/${R}_pmc_$c.json
done
[ -s $T/configs_to_termination.jsonl ] && cp $T/configs_to_termination.jsonl profiles/${R}_configs_to_termination.jsonl
ls -la profiles/${R}_* | awk '{print $5, $9}'
