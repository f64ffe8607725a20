#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, separate runs) -> per-kernel HBM-side traffic per launch.

usage: pmc_hbm.py <fetch_dir> <write_dir> <out_all.json> <out_dominant.json> [calibration.json from pmc_calib.py]
The second file maps bench.py's per-launch event names (mlp_fused[C=..], attn_fused[C=..]) to bytes per launch.
"""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

CP_TO_C = {48: 45, 80: 72, 96: 96, 144: 144, 192: 192, 384: 384}


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name).replace("escx::", ""))


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[short(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


CODEC_KERNEL_SOURCES = ("fused_attn.h", "fused_deembed.h", "fused_mlp.h", "fused_rowgemm.h", "fused_swin.hip", "gemm_engine.h", "gemm_misc.hip",
                        "gemm_swin.hip", "kernels.h", "kernels_misc.hip", "launchers.h", "fused_pvq.h", "tune_env.h", "fused_mlp_x3.h", "split_terms.h")


def csrc_hash():
    """sha256 over the sources of the kernels the codec bench launches (the training / discriminator / collective files and the host launch
    sequence do not change what those kernels read or write): bench.py refuses a traffic figure measured on other kernels."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "efficient-speech-codec_amd", "csrc")
    h = hashlib.sha256()
    for f in CODEC_KERNEL_SOURCES:
        h.update(f.encode()); h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


def main():
    fd, wd, out_all, out_dom = sys.argv[1:5]
    cal = json.load(open(sys.argv[5])) if len(sys.argv) > 5 and os.path.exists(sys.argv[5]) else {}
    ff, wf = cal.get("fetch_factor") or 1.0, cal.get("write_factor") or 1.0
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    allk, dom = {}, {}
    for k in sorted(set(fe) | set(wr)):
        n = max(fe[k][0], wr[k][0], 1)
        f = fe[k][1] / max(fe[k][0], 1); w = wr[k][1] / max(wr[k][0], 1)
        allk[k] = {"launches": n, "fetch_KiB_per_launch": round(f, 1), "write_KiB_per_launch": round(w, 1)}
        m = re.match(r"(mlp_fused_lds_kernel|mlp_x3_kernel|attn_fused_kernel|attn_packed_kernel)<(\d+),", k)
        if m:
            fam = "mlp_fused" if m.group(1) == "mlp_fused_lds_kernel" else ("mlp_x3" if m.group(1) == "mlp_x3_kernel" else "attn_fused")
            if fam == "mlp_x3" and re.search(r", true>", k):
                fam = "mlp_x3_split"                     # the PatchSplit-epilogue instantiations are their own launch group in bench.py
            ev = fam + f"[C={CP_TO_C.get(int(m.group(2)), int(m.group(2)))}]"
            tot = dom.setdefault(ev, [0, 0.0])
            tot[0] += n; tot[1] += n * (f * ff + w * wf) * 1024
    dom = {k: int(v[1] / v[0]) for k, v in dom.items()}
    dom["_note"] = ("bytes per launch = (FETCH_SIZE * fetch_factor + WRITE_SIZE * write_factor) KiB * 1024 from separate rocprofv3 --pmc passes of bench.py "
                    "(2 streams: each launch covers 18 of the 36 clips), launch-weighted over the kernel variants behind one event name; the factors come "
                    "from tools/pmc_calib.py (a copy with this library's access pattern over 1.5 GiB: true bytes / reported bytes).")
    dom["_calibration"] = {"fetch_factor": ff, "write_factor": wf}
    dom["_csrc_sha256"] = csrc_hash()
    json.dump(allk, open(out_all, "w"), indent=1)
    json.dump(dom, open(out_dom, "w"), indent=1)
    print(json.dumps(dom, indent=1))


if __name__ == "__main__":
    main()
