# the candidate update of the persistent candidate stage: product (dbg 0) against no update at all (measurement build, DBG 8: timing only),
# in-step launch (1024 queries) and C3 batch (100k), alternating processes.  (Rounds' logs: r06_k / r06_r used DBG 32 = unseeded / unshared,
# forms that left with the packed lists.)   needs the measurement library on the box: empty .gpurunignore for this call
export CSLAM_HIP_LIB=$PWD/cslam_amd/libcslam_hip_abl.so
for rep in 1 2; do for d in 0 8; do echo "dbg $d"; python tools/perf_match_ring.py 1024,100000 0 $d 5 2>&1 | grep "^nq\|Error"; done; done
