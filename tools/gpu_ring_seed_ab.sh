# the seeded first tile of the persistent candidate stage against the unseeded form (measurement build, DBG 32), alternating processes
export CSLAM_HIP_LIB=$PWD/cslam_amd/libcslam_hip_abl.so
for rep in 1 2; do for d in 0 32; do python tools/perf_match_ring.py 1024,100000 0 $d 5 2>&1 | grep "^nq"; done; done
