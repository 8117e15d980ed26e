cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/ft_events.py 2>&1 | grep -v "it/s" | tee gpurun_out/r6_ft_events.txt
