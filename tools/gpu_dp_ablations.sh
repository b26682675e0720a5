# timing-only ablations of csrc/conv_direct_p.hip (measurement build): bash tools/gpu_dp_ablations.sh
for n in 8 4; do for d in 0 1 2 4 16 7 23; do echo "NRW $n DBG $d"; CSLAM_HIP_LIB=$PWD/cslam_amd/libcslam_hip_abl.so CSLAM_DP_NRW=$n CSLAM_DP_DBG=$d timeout 120 python tools/perf_direct_p.py 1000 2>&1 | grep shortcut | sed -n '2p;5p'; done; done
