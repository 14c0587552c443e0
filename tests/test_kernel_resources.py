"""Register / spill budget of the hot kernels, read from the compiler's own metadata (hipcc -S cross-compiles for
gfx950 without a GPU).  A feature that inlines into a tree kernel can silently double its registers -- round 2's
Dirichlet sampler took k_tree_step from 78 to 129 VGPRs with SGPR spills until it got a kernel of its own -- so the
budgets are pinned here."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _resources(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only",
                           "-S", "-o", str(out), os.path.join(ROOT, "betazero_amd", "csrc", src)],
                          stderr=subprocess.DEVNULL)
    res = {}
    txt = out.read_text()
    _resources.asm = txt
    for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if name:
            g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))  # noqa: E731
            res[name.group(1)] = {"vgpr": g("vgpr_count"), "vspill": g("vgpr_spill_count"), "sspill": g("sgpr_spill_count"),
                                  "scratch": g("private_segment_fixed_size")}
    return res


def _find(res, *parts):
    hits = [k for k in res if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return res[hits[0]]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_tree_kernels_stay_within_their_register_budget(tmp_path):
    res = _resources("bz_mcts.hip", tmp_path)
    step = _find(res, "k_tree_step", "ReversiTILi8")          # cfg 3's tree step: 4+ waves per SIMD
    assert step["vgpr"] <= 96 and step["vspill"] == 0 and step["sspill"] == 0 and step["scratch"] == 0, step
    for gw, cap in (("ILi2E", 128), ("ILi4E", 128)):          # cfg 2's fused search (4 lanes per game is the default since round 3)
        for uni in ("Lb1E", "Lb0E"):                           # uniform evaluator (cfg 2) / hash evaluator
            k = _find(res, "k_search_fused_ttt", gw + uni)
            assert k["vgpr"] <= cap and k["vspill"] == 0 and k["sspill"] == 0 and k["scratch"] == 0, (gw, uni, k)
    play = _find(res, "k_play", "ReversiTILi8")
    assert play["vspill"] == 0 and play["scratch"] == 0, play


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_net_kernels_do_not_spill(tmp_path):
    res = _resources("bz_net.hip", tmp_path)
    for parts in (("k_tower_bf16", "Li128ELi4ELb1E"), ("k_tower_bf16", "Li64ELi8E"), ("k_tower_bf16", "Li256ELi2E"),
                  ("k_tower_bf16", "Li128ELi1E"), ("k_tower_fp8",)):
        k = _find(res, *parts)
        assert k["vspill"] == 0 and k["scratch"] == 0 and k["vgpr"] <= 512, (parts, k)
    assert _find(res, "k_tower_fp8")["vgpr"] <= 256  # two workgroups per CU


def _count(asm, kernel_parts, mnemonic):
    """occurrences of an instruction in one kernel's body"""
    for m in re.finditer(r"^(\S+):\s*; @\1$", asm, re.M):
        if all(p in m.group(1) for p in kernel_parts):
            body = asm[m.end():asm.index("s_endpgm", m.end())]
            return len(re.findall(rf"^\s+{mnemonic}\b", body, re.M))
    raise AssertionError(kernel_parts)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_towers_skip_the_padding_row_mfmas(tmp_path):
    """Row-tile units: per conv layer a wave issues 9 taps x 8 k-steps x 8 units = 576 MFMAs minus the 6
    taps x 8 k-steps whose unit reads only padding rows = 528 (bf16; fp8: 144 - 12 = 132).  Two layer bodies per kernel
    + stem + heads: a regression to position-major units would show up here as 1184 / 292."""
    _resources("bz_net.hip", tmp_path)
    asm = _resources.asm
    # round 4: the benchmark net's throughput shape runs on v_mfma_f32_16x16x32_bf16 (half the MACs each): 2 x 528 per
    # layer body, 2 x 16 for the stem (4 quarters x 8 units); the heads stay on 32x32x16 (16)
    assert _count(asm, ("k_tower_bf16", "Li128ELi4ELb1E"), "v_mfma_f32_16x16x32_bf16") == 2 * (2 * 528) + 32
    assert _count(asm, ("k_tower_bf16", "Li128ELi4ELb1E"), "v_mfma_f32_32x32x16_bf16") == 16
    assert _count(asm, ("k_tower_bf16", "Li64ELi8ELb0E"), "v_mfma_f32_32x32x16_bf16") == 2 * 264 + 8 + 16  # C = 64: 4 k-steps per tap
    assert _count(asm, ("k_tower_fp8",), "v_mfma_scale_f32_32x32x64_f8f6f4") == 2 * 132 + 4


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_training_kernels_do_not_spill(tmp_path):
    """csrc/bz_train.hip: forward / backward-data keep the inference tower's budget (one workgroup per CU), the
    weight-gradient kernel keeps its staging registers in registers (an array of HIP's uint4 STRUCT had put them into
    scratch memory: 272 bytes per lane) and really reads both operands through the LDS transpose read"""
    res = _resources("bz_train.hip", tmp_path)
    for parts in (("k_train_fwd", "Li64ELi8E"), ("k_train_fwd", "Li128ELi4ELb1E"), ("k_train_bwd", "Li64ELi8E"), ("k_train_bwd", "Li128ELi4ELb1E"),
                  ("k_train_fwd", "Li64ELi4E"), ("k_train_bwd", "Li64ELi4E"),
                  ("k_train_wgrad", "Li64E"), ("k_train_wgrad", "Li128E")):
        k = _find(res, *parts)
        assert k["vspill"] == 0 and k["sspill"] == 0 and k["scratch"] == 0 and k["vgpr"] <= 512, (parts, k)
    assert _count(_resources.asm, ("k_train_wgrad", "Li128E"), "ds_read_b64_tr_b16") >= 14


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_training_end_kernels_do_not_spill(tmp_path):
    """csrc/bz_train_ends.hip: the stem keeps its 8 x 18 weights, the stem's weight gradient its 2 x 19 and the FC
    weight-gradient kernel its 49 accumulators in registers (no scratch); the heads kernel holds a whole position's x tile
    in flight (64 registers at 128 channels) and stays far below what its one workgroup per CU (130 KB of LDS) may use"""
    res = _resources("bz_train_ends.hip", tmp_path)
    for parts in (("k_train_stemILi64E",), ("k_train_stemILi128E",), ("k_train_stem_wgradILi64E",), ("k_train_stem_wgradILi128E",),
                  ("k_train_headsILi64E",), ("k_train_headsILi128E",), ("k_train_heads_wgrad",), ("k_train_finish",), ("k_train_adam",)):
        k = _find(res, *parts)
        assert k["vspill"] == 0 and k["sspill"] == 0 and k["scratch"] == 0 and k["vgpr"] <= 512, (parts, k)
    assert _find(res, "k_train_headsILi128E")["vgpr"] <= 256
