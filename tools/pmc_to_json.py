"""profiles/<tag>_blend_pmc.txt / <tag>_attention_pmc.txt / <tag>_gemm_pmc.txt (tools/collect_<tag>.sh) -> the JSON files bench.py quotes, each with the SHA-256 of the
kernel source it was measured on: bench.py drops the quotation when the source has changed since.
usage: python tools/pmc_to_json.py [tag = r6]   (run in the repo root after copying the two text files into profiles/)"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"      # round tag of the text files: profiles/<TAG>_{blend,attention,gemm}_pmc.txt


def sha(rel):
    return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()


def counters(path, kernel_prefix):
    """{kernel line: {counter: avg}} for the blocks whose kernel name starts with kernel_prefix"""
    out, cur = {}, None
    for line in open(path):
        if line.startswith("#"):
            continue
        m = re.match(r"\s+(\w+)\s+avg\s+([0-9.]+)", line)
        if m and cur is not None:
            out[cur][m.group(1)] = float(m.group(2))
        elif line.strip():
            cur = line.strip() if kernel_prefix in line else None
            if cur is not None:
                out.setdefault(cur, {})
    return out


def main():
    b = {}
    for c in counters(os.path.join(ROOT, "profiles", TAG + "_blend_pmc.txt"), "surfel_blend_kernel<false>").values():
        b.update(c)
    fetch, write = b["FETCH_SIZE"], b["WRITE_SIZE"]
    gui = b["GRBM_GUI_ACTIVE"]
    blend = {
        "_comment": "surfel_blend_kernel<false> per launch at BASELINE configs[1], rocprofv3 PMC passes (profiles/" + TAG + "_blend_pmc.txt, separate --pmc "
                    "passes with --kernel-trace only; tools/collect_" + TAG + ".sh).  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE under-reports "
                    "wide (16 B / lane) reads by 2x (MI355X_MICROARCH.md, HBM section), so the fetch is doubled.  valu_issue_frac = "
                    "SQ_INSTS_VALU x 4 cycles / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs.",
        "workload": "surface scene, 100000 surfels x 8 views x 512x512",
        "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
        "traffic_bytes_per_launch": int((2 * fetch + write) * 1024),
        "SQ_INSTS_VALU": b["SQ_INSTS_VALU"], "SQ_INSTS_SALU": b.get("SQ_INSTS_SALU"), "SQ_INSTS_LDS": b.get("SQ_INSTS_LDS"),
        "GRBM_GUI_ACTIVE": gui, "SQ_LDS_IDX_ACTIVE": b.get("SQ_LDS_IDX_ACTIVE"), "SQ_LDS_BANK_CONFLICT": b.get("SQ_LDS_BANK_CONFLICT"),
        "valu_issue_frac": round(b["SQ_INSTS_VALU"] * 4 / 1024 / (gui / 8), 4),
        "kernel_cycles": gui / 8,
        "source": "profiles/" + TAG + "_blend_pmc.txt",
        "source_sha256": {rel: sha(rel) for rel in ("gaussiananything_amd/csrc/surfel_blend.hip", "gaussiananything_amd/csrc/surfel_common.h",
                                                    "gaussiananything_amd/csrc/Makefile")},   # (the build flags count: see the Makefile)
    }
    json.dump(blend, open(os.path.join(ROOT, "profiles", TAG + "_blend_pmc.json"), "w"), indent=1)
    att = {}
    for name, c in counters(os.path.join(ROOT, "profiles", TAG + "_attention_pmc.txt"), "attention_fwd_kernel").items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            att[name[:60]] = {"mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 4),
                              "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"]}
    json.dump({"_comment": "MFMA busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs) of the two attention launches of "
                           "bench.py's `attention` section (tools/dit_kernels_two.py attn; profiles/" + TAG + "_attention_pmc.txt)",
               "kernels": att, "source": "profiles/" + TAG + "_attention_pmc.txt",
               "source_sha256": {rel: sha(rel) for rel in ("gaussiananything_amd/csrc/dit_attention.hip", "gaussiananything_amd/csrc/dit_common.h",
                                                         "gaussiananything_amd/csrc/Makefile")}},
              open(os.path.join(ROOT, "profiles", TAG + "_attention_pmc.json"), "w"), indent=1)
    gm = {}
    for name, c in counters(os.path.join(ROOT, "profiles", TAG + "_gemm_pmc.txt"), "gemm_").items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            gm[name[:60]] = {"mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 4),
                             "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"]}
    json.dump({"_comment": "MFMA busy of the GEMM launches of a DiT-L block (tools/dit_kernels_two.py gemm, cold weights; profiles/" + TAG + "_gemm_pmc.txt): "
                           "<0,4,2,3,4,4,2> qkv 1536x3072x1024, <1,4,2,3,4,4,2> fc1 1536x4096x1024 (GELU), <2,2,2,3,2,4,0> fc2 1536x1024x4096 and proj "
                           "1536x1024x1024 (residual), <0,4,1,1,4,4,0> the 768x1024x1024 cross-attention projection shape",
               "kernels": gm, "source": "profiles/" + TAG + "_gemm_pmc.txt",
               "source_sha256": {rel: sha(rel) for rel in ("gaussiananything_amd/csrc/dit_gemm.hip", "gaussiananything_amd/csrc/dit_common.h",
                                                         "gaussiananything_amd/csrc/Makefile")}},
              open(os.path.join(ROOT, "profiles", TAG + "_gemm_pmc.json"), "w"), indent=1)
    print(json.dumps(gm, indent=1))
    print(json.dumps(blend, indent=1)[:600])
    print(json.dumps(att, indent=1))


if __name__ == "__main__":
    main()
