cd $GRAFT_REPO_ROOT
python tools/latency_ablate.py base:coop=0,ch3:coop=0 --sizes 2048,4096,8192,16384
python tools/ab_bench.py --names base,ch3 --rounds 3 --steps 1280 2>&1 | tail -4
python tools/ab_bench.py --names base,ch3 --rounds 2 --steps 20 --group 20 --extra "--warmup 5" 2>&1 | tail -3
