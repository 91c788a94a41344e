# parity margins of every fixture in every mode + PMC passes of the final build (GPU box)
mkdir -p gpurun_out/r03e
python3 profiles/parity_margins.py gpurun_out/r03e/parity_margins.json > gpurun_out/r03e/parity_margins.log 2> gpurun_out/r03e/parity_margins.err; tail -2 gpurun_out/r03e/parity_margins.log
bash profiles/collect_pmc.sh r03e_pmc && python3 profiles/pmc_to_json.py r03e_pmc gpurun_out/r03e/pmc_k_row_tasks.json
du -sh gpurun_out | tail -1
