cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for a in 0 4 1 5; do echo "== ablate $a"; LEMON_PHASE_PROF=1 LEMON_ABLATE=$a timeout -k 10 300 python3 tools/scan_time.py 50000 40000 512 51 f32 2>&1 | grep -E "phase|lives|best" ; done
