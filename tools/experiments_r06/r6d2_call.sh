tools/experiments_r06/pmc_fetch.sh base variants/libfsr1_base.so --math f 2>&1 | tee gpurun_out/r6d2_pmc_rcas_updown.log
tools/experiments_r06/pmc_fetch.sh updown "" --math f 2>&1 | tee -a gpurun_out/r6d2_pmc_rcas_updown.log
