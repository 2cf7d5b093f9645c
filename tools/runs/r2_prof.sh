export GPU_MAX_HW_QUEUES=8
bash tools/pmc_extractor.sh > gpurun_out/r02_pmc_extractor.txt 2>&1
tail -40 gpurun_out/r02_pmc_extractor.txt
bash tools/profile.sh > gpurun_out/r02_profile_log.txt 2>&1
tail -60 gpurun_out/profile_summary.txt
