#!/usr/bin/env python
"""Roofline calibration on the box at hand: the sustained dense-f16 MFMA rate with every SIMD running nothing but
independent v_mfma_f32_32x32x16_f16 chains (rcdm_debug_mfma_peak), the shader clock that rate implies
(1024 SIMDs x 1024 flop/clk), and the tick rate of s_memtime (the unit of tools/trace_igemm.py / trace_phases.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip
lib = hip.load()
dev = torch.device("cuda", 0)
cus = torch.cuda.get_device_properties(0).multi_processor_count
for waves_per_simd in (1, 2):
    blocks = cus * waves_per_simd
    ticks = torch.zeros(blocks, dtype=torch.int64, device=dev)
    sink = torch.zeros(1, dtype=torch.float32, device=dev)
    for iters, reps in ((20000, 1), (200000, 3)):
        best = None
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.rcdm_debug_mfma_peak(blocks, iters, sink.data_ptr(), ticks.data_ptr(), hip.stream_ptr())
            assert rc == 0, rc
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        flop = blocks * 4 * iters * 4 * 32768.0
        tf = flop / (best * 1e-3) / 1e12
        tk = float(ticks.double().median())
        print(f"{waves_per_simd} wave(s)/SIMD, {iters} x 4 MFMA per wave: {best:8.3f} ms -> {tf:7.1f} TFLOP/s dense f16 "
              f"= {tf * 1e12 / (4 * cus * 1024) / 1e9:.3f} GHz x {4 * cus} SIMDs x 1024 flop/clk; "
              f"s_memtime {tk / (best * 1e-3) / 1e6:.1f} MHz")
