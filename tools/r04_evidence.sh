# Round-4 evidence visit (stamped with lib/build_stamp.json): kernel traces + PMC passes of the final build.
# usage: bash tools/r04_evidence.sh <part>   (parts: trace | pmc16 | pmcx3 | cliff)
set -u
case ${1:-trace} in
  trace)  SKIP_TESTS=1 bash tools/gpu_check.sh r04_final && SKIP_TESTS=1 WORKLOAD=xlsr_aasist bash tools/gpu_check.sh r04_final_teacher;;
  pmc16)  bash tools/pmc_traffic.sh r04_pmc_traffic && bash tools/pmc_mfma.sh r04_pmc_mfma && AFX_WORKLOAD=xlsr_aasist bash tools/pmc_traffic.sh r04_pmc_traffic_teacher && AFX_WORKLOAD=xlsr_aasist bash tools/pmc_mfma.sh r04_pmc_mfma_teacher;;
  pmcx3)  export AFX_DTYPE=fp16x3; bash tools/pmc_traffic.sh r04_pmc_traffic_fp16x3 && bash tools/pmc_mfma.sh r04_pmc_mfma_fp16x3 && AFX_WORKLOAD=xlsr_aasist bash tools/pmc_traffic.sh r04_pmc_traffic_teacher_fp16x3 && AFX_WORKLOAD=xlsr_aasist bash tools/pmc_mfma.sh r04_pmc_mfma_teacher_fp16x3;;
  cliff)  bash tools/diag_queue_cliff.sh r04_queue_cliff;;
esac
