# where should the small-tile path hand over to the large-batch kernels?  us per denoise step for several batches
for lim in 0 1024 2048 4096 8192; do echo "MST_SMALL_M=$lim"; MST_SMALL_M=$lim HB_BATCHES=1,2,4,8,16,32 timeout -k 10 200 python tools/host_bound.py 2>/dev/null; done
