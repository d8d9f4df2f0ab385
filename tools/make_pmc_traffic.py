"""profiles/pmc_traffic.json from the round's PMC summaries (gpurun_out/r03_pmc_*_{fetch,write,mfma}.txt, written by
tools/collect_profiles.sh): fabric bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE reports half of a
wide coalesced read, MI355X_MICROARCH.md "HBM"), and the MFMA pipe's busy cycles per SIMD per launch.
usage: python tools/make_pmc_traffic.py <dir with the r03_pmc_*.txt> <date>"""
import json, os, re, sys
d, date = sys.argv[1], sys.argv[2]


def means(tag, group):
    out, k = {}, None
    for line in open(os.path.join(d, "r03_pmc_%s_%s.txt" % (tag, group))):
        if not line.startswith(" "):
            k = line.strip()
            out[k] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
            out[k][m.group(1)] = float(m.group(3))
    return out


def pick(ms, needle):
    hits = [v for k, v in ms.items() if needle in k]
    assert len(hits) == 1, (needle, list(ms))
    return hits[0]


def entry(tag, needle, what, mfma=False):
    f = pick(means(tag, "fetch"), needle)["FETCH_SIZE"]
    w = pick(means(tag, "write"), needle)["WRITE_SIZE"]
    e = {"traffic_bytes": int(round((2 * f + w) * 1024)),
         "source": "profiles/r03_pmc_%s_fetch.txt / _write.txt (%s: FETCH_SIZE %.0f KB x2 gfx950 correction + WRITE_SIZE %.0f KB)" % (tag, what, f, w),
         "date": date}
    if mfma:
        m = pick(means(tag, "mfma"), needle)
        e["mfma_busy_cycles_per_simd"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0, 1)
        e["mfma_insts"] = int(m["SQ_INSTS_MFMA"])
        e["mfma_source"] = "profiles/r03_pmc_%s_mfma.txt (SQ_VALU_MFMA_BUSY_CYCLES %.4g = 16.0 x SQ_INSTS_MFMA, over 1024 SIMDs)" % (tag, m["SQ_VALU_MFMA_BUSY_CYCLES"])
    return e


out = {
    "layer2_bf16": entry("dense_l2", "k_dense_bf16<4, 5, 2, 2, 4, 1,", "k_dense_bf16<4,5,2,2,4> bf16 out, layer-2 shape alone", mfma=True),
    "layer2_head_bf16": entry("step", "k_dense_bf16<4, 5, 2, 2, 4, 3,", "k_dense_bf16<4,5,2,2,4,YM=3>: layer 2 fused with the head, in the step", mfma=True),
    "layer1_bf16": entry("step", "k_dense_bf16<4, 5, 2, 2, 4, 1,", "k_dense_bf16<4,5,2,2,4> bf16 out: layer 1 in the step", mfma=True),
    "draw_multi": entry("step", "k_draw_multi", "k_draw_multi: 6 tensors x 8 samples, the KL partial sums from its own items"),
    "conv_lenet_bf16": entry("conv_lenet", "k_conv_bf16", "k_conv_bf16<4,4>"),
    "conv_cifar_bf16": entry("conv_cifar", "k_conv_bf16", "k_conv_bf16<8,6>"),
}
wf, ww = means("wide", "fetch"), means("wide", "write")
tot = sum((2 * pick(wf, n)["FETCH_SIZE"] + pick(ww, n)["WRITE_SIZE"]) * 1024 for n in ("k_dense_bf16", "k_draw_multi", "k_split_bf16x3"))
out["wide_f32"] = {"traffic_bytes": int(round(tot)), "date": date,
                   "source": "profiles/r03_pmc_wide_fetch.txt / _write.txt (the layer call's three launches: k_dense_bf16<4,8,4,1,3> on three-plane operands "
                             "%.0f MB + k_draw_multi (three planes) %.0f MB + k_split_bf16x3 %.0f MB)" % tuple(
                                 (2 * pick(wf, n)["FETCH_SIZE"] + pick(ww, n)["WRITE_SIZE"]) * 1024 / 1e6 for n in ("k_dense_bf16", "k_draw_multi", "k_split_bf16x3"))}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
