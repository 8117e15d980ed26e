cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/ft_host_profile.py 2>&1 | grep -v "it/s" > gpurun_out/r6_ft_host_profile.txt; tail -80 gpurun_out/r6_ft_host_profile.txt
