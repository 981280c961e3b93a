"""The parts of bench.py (repo root): `python bench.py --gpus N --steps K --warmup W` is the driver's contract and stays a thin
argument parser + orchestration; what it runs lives here, one concern per module:

    launch.py    arguments, `python bench.py --gpus N` starting its own ranks, the process group
    step.py      the TIMED REGION and nothing else: the workload resident in HBM, one step, the timed loop
    multi.py     N > 1: every mode of the sinogram assembly measured, the mode of the timed loop chosen from them
    modes.py     what is reported beside `value`: the Newton modes against the exact count, the noisy step, reduced
                 quadrature, single row, cone beam, mixed precision
    roofline.py  the `roofline` / `roofline_siddon` objects
    dropin.py    the public NumPy boundary, wall-clock
    cpu.py       `cpu_baseline` + the parity sample - the only module that touches oracle/
"""
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64
CLOCK_GHZ = 2.4                # MI355X_MICROARCH.md: max clock (the traversal kernel holds ~2.2 under load)
SIMDS = 1024                   # 256 CUs x 4
ISSUE_PEAK_G = SIMDS * CLOCK_GHZ / 4.0      # G wave-instructions/s the chip can issue (one per SIMD every 4 cycles): 614.4
