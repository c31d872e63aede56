"""Clock and matrix-pipe occupancy of the conv kernels on one layer shape, from a rocprofv3 PMC pass (tools/README.md):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d OUT -o run -- \
        python3 tools/probe_clock.py run
    python3 tools/probe_clock.py report OUT/run_counter_collection.csv

`run-wino B` (round 4): the fp32 default routing -- Winograd-domain grouped products, the transform sweeps and the grouped
weight-grad -- on the step's 3x3 layers at per-GPU batch B; a second pass with --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY adds the wait shares to the report.
`run` launches forward, data-grad and weight-grad of one SAME 3x3 conv through ops.* (the product path) in fp32, bf16 and
fp8 mode after ~1.5 s of back-to-back warm-up launches per mode (the chip lowers its clock under matrix load).
`report`: effective clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch wall time (MI355X_MICROARCH.md 'DVFS give-back');
matrix-pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles of the dispatch)."""
import csv
import math
import os
import sys
import time

csv.field_size_limit(sys.maxsize)


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from scrabble_gan_amd import ops
    dev = torch.device("cuda:0")
    B, H, W, Cin, Cout, k = 128, 16, 80, 512, 512, 3
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(k, k, Cin, Cout, device=dev, generator=g) / math.sqrt(k * k * Cin)
    dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
    y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.zeros_like(w)
    for mode in ("f32-gen1", "f32", "bf16", "fp8"):
        ops.USE_F32_V2 = mode != "f32-gen1"
        ops.set_conv_dtype(mode.split("-")[0])

        def step():
            ops.new_step()
            ops.conv2d_fwd(x, w, None, relu_in=True, out=y)
            ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=dx)
            ops.conv2d_bwd_weight(x, dy, dw, relu_in=True)
        t0 = time.time()
        while time.time() - t0 < 1.5:
            step()
            torch.cuda.synchronize()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    ops.set_conv_dtype("f32")


WINO_LAYERS = [(16, 80, 512, 512), (8, 40, 1024, 1024), (4, 20, 1024, 1024), (16, 80, 64, 512), (8, 80, 256, 256), (16, 160, 128, 128)]


def run_wino(B):
    """fp32 mode with the default routing (Winograd-domain products, the transforms around them and the grouped weight-grad) on the
    3x3 layers of the step at per-GPU batch B -- the kernels that carry the c2 step since round 3."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from scrabble_gan_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    ops.set_conv_dtype("f32")
    for (H, W, Cin, Cout) in WINO_LAYERS:
        x = torch.randn(B, H, W, Cin, device=dev, generator=g)
        w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g) / math.sqrt(9 * Cin)
        dy = torch.randn(B, H, W, Cout, device=dev, generator=g)
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.zeros_like(w)

        def step():
            ops.new_step()
            ops.conv2d_fwd(x, w, None, relu_in=True, out=y)
            ops.conv2d_bwd_data(dy, w, (H, W), mask=x, out=dx)
            ops.conv2d_bwd_weight(x, dy, dw, relu_in=True)
        t0 = time.time()
        while time.time() - t0 < 1.0:
            step()
            torch.cuda.synchronize()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        del x, w, dy, y, dx, dw


def report(path):
    rows = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            if "sg_igemm" not in name and "sg_wgrad" not in name and "k_w43" not in name and "k_wino" not in name:
                continue
            d = rows.setdefault(r["Dispatch_Id"], {"name": name, "grid": r["Grid_Size"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # the last 9 dispatches of each kernel name+grid are the timed ones; print the median by wall time of each group
    groups = {}
    for d in rows.values():
        groups.setdefault((d["name"], d["grid"]), []).append(d)
    print("%-64s %9s %9s %10s %9s" % ("kernel (grid)", "wall ms", "clock GHz", "MFMA busy", "launches"))
    for (name, grid), ds in groups.items():
        ds = sorted(ds[-3:], key=lambda d: d["ns"])
        d = ds[len(ds) // 2]
        cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        clk = cyc / d["ns"] if d["ns"] else 0.0
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else 0.0
        extra = ""
        if d.get("SQ_WAVE_CYCLES"):          # second pass (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY): share of the resident wave-cycles
            wc = d["SQ_WAVE_CYCLES"]
            extra = "  waiting %4.1f%%  waiting on an instruction result %4.1f%%  issuing %4.1f%%" % (
                100.0 * d.get("SQ_WAIT_ANY", 0.0) / wc, 100.0 * d.get("SQ_WAIT_INST_ANY", 0.0) / wc, 100.0 * d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc)
        print("%-64s %9.3f %9.3f %9.1f%% %9d%s" % ((name + " (" + grid + ")")[:64], d["ns"] * 1e-6, clk, 100.0 * busy, len(groups[(name, grid)]), extra))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    elif len(sys.argv) >= 3 and sys.argv[1] == "run-wino":
        run_wino(int(sys.argv[2]))
    else:
        run()
