mkdir -p gpurun_out/r03c
python -m pytest tests -x -q -m gpu > gpurun_out/r03c/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03c/gputests.log; tail -4 gpurun_out/r03c/gputests.log
timeout -k 10 400 python tools/part_profile.py --world 8 --workload cfg3_netflix --scaling strong > gpurun_out/r03c/pp8_netflix_strong.log 2>&1
grep -h "sum\|plan\|part" gpurun_out/r03c/pp8_netflix_strong.log | cut -c1-260
