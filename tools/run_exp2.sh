cd /root/repo
bash tools/kt.sh s1m --workload synth1M
RT_AMD_DEBUG=1 timeout -k 10 200 python tools/prof_frame.py --frames 1 --workload synth1M --flags 4096 2>&1 | grep "level\|beam traversal\|listed pairs"
