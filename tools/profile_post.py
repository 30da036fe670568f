#!/usr/bin/env python3
"""Turns the outputs of tools/profile_round.sh (gpurun_out/<tag>_*) into the committed evidence under profiles/:
copies the kernel-stats / PMC summaries / bench line and derives profiles/<tag>_roofline_inputs.json -- per-launch HBM bytes
of the dominant kernel (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes, KiB per dispatch) and the in-step average
durations + GB/s of the HBM-bound kernels -- which bench.py reports as roofline.traffic / roofline.hbm_kernels.
    python tools/profile_post.py r02
The kernel-trace / PMC passes run bench.py with --no-probe, so the in-step average of the dominant kernel counts the step's own
launches only (round 3's folded the roofline probe's 50 launches in)."""
import json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, N, D = 64, 196, 512


def kernel_stats(path):
    rows = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            rows[m.group(1).strip()] = dict(calls=int(m.group(2)), avg_us=float(m.group(4)))
    return rows


def pmc(path):
    lines = open(path).read().split("\n")
    names = lines[0].split()[5:] if lines else []
    out = {}
    for ln in lines[1:]:
        if not ln.strip():
            continue
        key = ln[:80].strip()
        vals = ln[80:].split()
        out[key] = vals
    return lines[0], out


def main(tag):
    O = os.path.join(ROOT, "gpurun_out")
    P = os.path.join(ROOT, "profiles")
    for suffix in ("kernel_stats.txt", "pmc_hbm.txt", "pmc_sq.txt", "pmc_lds_valu.txt", "bench.json"):
        src = os.path.join(O, "%s_%s" % (tag, suffix))
        if os.path.exists(src):
            shutil.copy(src, os.path.join(P, "%s_%s" % (tag, suffix)))
    ks = kernel_stats(os.path.join(O, "%s_kernel_stats.txt" % tag))
    head, hb = pmc(os.path.join(O, "%s_pmc_hbm.txt" % tag))
    cols = head.split()[4:]            # counter names in the header after "kernel (mean per dispatch)"
    fi = [i for i, c in enumerate(cols) if "FETCH" in c][0]
    wi = [i for i, c in enumerate(cols) if "WRITE" in c][0]

    def find(d, pat):
        for k in d:
            if re.search(pat, k):
                return k, d[k]
        return None, None

    gk, gv = find(hb, r"chain_fwd_kernel<512")
    if not gv:
        gk, gv = find(hb, r"kb_gemm_h2_kernel<13, 0, 0, false>")
    out = {"source": "tools/profile_round.sh %s: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of bench.py "
                     "(KiB per dispatch; FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md HBM section), and --kernel-trace --stats" % tag}
    if gv:
        fetch, write = float(gv[fi]), float(gv[wi])
        out.update(kernel=gk, FETCH_SIZE_KiB=fetch, WRITE_SIZE_KiB=write, hbm_bytes_per_launch=int((2 * fetch + write) * 1024))
    sk, sv = find(ks, r"chain_fwd_kernelILi512ELi\d+ELi64")
    if not sv:
        sk, sv = find(ks, r"kb_gemm_h2_kernelILi13ELi0ELi0ELb0")
    if sv:
        out["in_step_kernel_ms"] = round(sv["avg_us"] / 1e3, 5)
    elem = B * N * D
    hbm = []
    for pat, name, byts in (                            (r"h2_from_f32_kernel", "h2_from_f32_kernel (KB -> H2 through the read dropout, + keep bits/bytes)", elem * 8 + elem // 4),
                            (r"read_att_bwd_h2_kernel", "read_att_bwd_h2_kernel (I2 H2 in, dI2 H2 out)", elem * 8),
                            (r"kb_attend_kernel", "kb_attend_kernel (softmax over N + sum_n a KB)", elem * 4),
                            (r"kb_att_da_kernel", "kb_att_da_kernel (da = dr . KB)", elem * 4)):
        k, v = find(ks, pat)
        if v:
            hbm.append({"kernel": name, "bound": "hbm", "bytes_per_launch": byts, "kernel_ms": round(v["avg_us"] / 1e3, 5),
                        "achieved": round(byts / (v["avg_us"] * 1e-6) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(byts / (v["avg_us"] * 1e-6) / 8e12, 4)})
    out["hbm_kernels"] = hbm
    out["file"] = "profiles/%s_roofline_inputs.json" % tag
    json.dump(out, open(os.path.join(P, "%s_roofline_inputs.json" % tag), "w"), indent=1)
    json.dump(out, open(os.path.join(P, "latest_roofline_inputs.json"), "w"), indent=1)      # what bench.py reads
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
