# Evidence of the final round-3 build, one GPU box: full -m gpu suite, smoke, rocprof kernel stats (cfg4, cfg5-small),
# the default bench line, parity margins of every fixture, PMC passes, other sizes.  Outputs under gpurun_out/r03z/.
mkdir -p gpurun_out/r03z
python -m pytest tests -m gpu -q > gpurun_out/r03z/gpu_tests_full.log 2>&1; tail -2 gpurun_out/r03z/gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash profiles/collect_r03.sh r03z_cfg4 && echo cfg4 stats done
bash profiles/collect_r03.sh r03z_cfg5s --size cfg5-small && echo cfg5s stats done
python3 bench.py > gpurun_out/r03z/bench_full.json 2> gpurun_out/r03z/bench_full.err; tail -c 300 gpurun_out/r03z/bench_full.json; echo
python3 profiles/parity_margins.py gpurun_out/r03z/parity_margins.json > gpurun_out/r03z/parity_margins.log 2> gpurun_out/r03z/parity_margins.err; tail -1 gpurun_out/r03z/parity_margins.log | cut -c1-200
bash profiles/collect_pmc.sh r03z_pmc && python3 profiles/pmc_to_json.py r03z_pmc gpurun_out/r03z/pmc_k_row_tasks.json
python3 profiles/iteration_host_profile.py cfg2 > gpurun_out/r03z/host_profile_cfg2.txt 2>&1; head -3 gpurun_out/r03z/host_profile_cfg2.txt | tail -1
python3 bench.py --size cfg5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03z/cfg5.json 2> gpurun_out/r03z/cfg5.err; tail -c 300 gpurun_out/r03z/cfg5.json; echo
python3 profiles/cond_estimates.py > gpurun_out/r03z/cond_estimates.txt 2> gpurun_out/r03z/cond_estimates.err; head -4 gpurun_out/r03z/cond_estimates.txt
du -sh gpurun_out | tail -1
