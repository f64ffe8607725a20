#!/usr/bin/env python
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access pattern.

    pmc_calib.py run                      # under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: 8 copies of a 1.5 GiB buffer of 192-byte rows
    pmc_calib.py reduce <fetch_dir> <write_dir> <out.json>     # factors = true bytes / reported bytes

The copy kernel (escx_test_copy_rows) reads and writes every byte exactly once with 16-byte accesses of 64-byte row segments, the pattern
of the fused MLP / attention kernels; the buffers are 6x the 256 MiB Infinity Cache, so nothing is served on-die.
"""
import csv, ctypes, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
ROWS, CP, REPS = 8 * 1024 * 1024, 48, 8          # 8 Mi rows x 192 B = 1.5 GiB


def run():
    import torch
    from esc import _native
    lib = _native.load()
    src = torch.ones(ROWS * CP, device="cuda"); dst = torch.empty_like(src)
    for _ in range(REPS):
        _native.check(lib.escx_test_copy_rows(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ROWS, CP, None))
    torch.cuda.synchronize()
    assert float(dst[-1]) == 1.0


def reduce(fd, wd, out):
    def per_launch(d, counter):
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter and "test_copy_rows" in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
        return sum(vals) / max(len(vals), 1), len(vals)
    true_bytes = ROWS * CP * 4
    f, nf = per_launch(fd, "FETCH_SIZE"); w, nw = per_launch(wd, "WRITE_SIZE")
    res = {"true_bytes_per_launch": true_bytes, "fetch_reported_KiB": f, "write_reported_KiB": w, "launches": [nf, nw],
           "fetch_factor": true_bytes / (f * 1024) if f else None, "write_factor": true_bytes / (w * 1024) if w else None,
           "pattern": f"{ROWS} rows x {CP * 4} B, 16 lanes per row, 16 B per lane (escx_test_copy_rows)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        reduce(*sys.argv[2:5])
