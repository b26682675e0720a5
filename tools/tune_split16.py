"""RESULT (torch 2.10 + ROCm 7.0): TunableOp does not intercept `bmm` with `out_dtype`, so this tunes nothing and the table
comes back unchanged (profiles/r01_exp_split16.log); kept for the torch release that does.
Intent: extend vpr/tunableop_gfx950.csv with the fp16 -> fp32 strided-batched GEMM shapes of the split-fp16 trunk layers
(VGG-16 at the 256-frame chunk: conv3_2/3_3, conv4_1, conv4_2/4_3, conv5_x).  Existing entries are kept (read first),
the table is rewritten after every shape so that a time-out loses only the shape in flight.  Run on the GPU box:
    python tools/tune_split16.py gpurun_out/tunableop_split16.csv"""
import os
import sys
import time
import torch
import torch.cuda.tunable as tunable

out = sys.argv[1]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(150)
tunable.set_max_tuning_iterations(30)
tunable.set_filename(out + '.torch')          # torch's own (incremental) writer; the table below is written by hand
tunable.read_file(os.path.join(here, "cslam_amd", "vpr", "tunableop_gfx950.csv"))
shapes = [(50176, 256, 256), (12544, 256, 512), (12544, 512, 512), (4096, 512, 512)]
for T, cin, cout in shapes:
    a = torch.randn(36, T, 3 * cin, device="cuda").half()
    b = torch.randn(36, 3 * cin, cout, device="cuda").half()
    t0 = time.perf_counter()
    torch.bmm(a, b, out_dtype=torch.float32)
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        torch.bmm(a, b, out_dtype=torch.float32)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"[36,{T},{3 * cin}] x [36,{3 * cin},{cout}]: tuned in {tt:.0f} s -> {ms:.3f} ms "
          f"({2.0 * 36 * T * 3 * cin * cout / ms / 1e9:.0f} TFLOP/s fp16)", flush=True)
    with open(out, "w") as f:
        for k, v in tunable.get_validators():
            f.write(f"Validator,{k},{v}\n")
        for r in tunable.get_results():
            f.write(",".join(str(c) for c in r) + "\n")
    del a, b
