"""Properties of the gfx950 code the product library ships, read from its code objects (no GPU needed): every kernel is free of
register spills and scratch, the hot kernels use the instructions DESIGN.md says they do, and the LDS-DMA weight-gradient kernel's
tile loop is not re-serialised by a compiler-inserted s_waitcnt vmcnt(0) (hipcc puts one between a builtin LDS-DMA and
the next transposing LDS read; the kernel issues its DMA from inline asm to keep two tiles in flight)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "4dflownet_amd", "lib4dflow_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump")]
    if not (os.path.exists(LIB) and all(os.path.exists(t) for t in tools)):
        pytest.skip("library or LLVM binutils not present")
    d = tmp_path_factory.mktemp("isa")
    fat = str(d / "fat.bin")
    subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, LIB, str(d / "copy.so")], check=True, capture_output=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundles in .hip_fatbin"
    out = []
    for i, s in enumerate(starts):                          # one bundle per translation unit
        part = str(d / ("bundle%d.bin" % i))
        open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = str(d / ("co%d.o" % i))
        subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part,
                        "--output=" + co], check=True, capture_output=True)
        notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
        asm = subprocess.run([tools[3], "-d", co], check=True, capture_output=True, text=True).stdout
        out.append((notes, asm))
    return out


def _kernels(notes):
    """{kernel name: {metadata key: int}} from the AMDGPU metadata note."""
    res, cur = {}, None
    block = {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and line.lstrip().startswith("-"):       # first key of a kernel entry
            block = {}
        if k == "name" and "kernel" in v:
            cur = v
            res[cur] = block
        if v.isdigit():
            block[k] = int(v)
    return res


def _function(asm, name_part):
    m = re.search(r"^[0-9a-f]+ <([^>]*%s[^>]*)>:\n(.*?)(?=^\S|\Z)" % re.escape(name_part), asm, re.S | re.M)
    return m.group(2) if m else None


def test_no_kernel_spills_or_uses_scratch(code_objects):
    seen = 0
    for notes, _ in code_objects:
        for name, md in _kernels(notes).items():
            seen += 1
            assert md.get("vgpr_spill_count", 0) == 0, (name, md)       # (SGPR spills go to VGPR lanes: no memory traffic)
            assert md.get("private_segment_fixed_size", 0) == 0, (name, md)
    assert seen >= 60, "expected the whole kernel set, saw %d" % seen


# Every kernel of the PRODUCT library and the include/fdn.h entry point that reaches it.  Kernels only a fdn_debug_* switch can select
# (the VALU thin-layer kernels, the cs4 direct layouts, the occupancy variants) are compiled into lib4dflow_hip_test.so only.
PRODUCT_KERNELS = {
    # conv64_wino2d_kernel<FUSED, HM, MB, SPLIT, MASK>, conv64_wino2d_shell_kernel<HM, MB, MASK>
    "conv64_wino2d_kernel<false, 4, 2, false, false>": "fdn_conv3d_fwd / fdn_conv3d_dgrad, 64->64, H % 4 == 0 and W % 4 == 0 (F(4,3) x F(4,3))",
    "conv64_wino2d_kernel<true, 4, 2, false, false>": "fdn_conv3d_dgrad_fused_part(FDN_DGRAD_INNER)",
    "conv64_wino2d_shell_kernel<4, 2, false>": "fdn_conv3d_dgrad_fused",
    "conv64_wino2d_kernel<false, 4, 1, false, false>": "... half-size tiles: grids whose full tiles leave CUs with one workgroup or none",
    "conv64_wino2d_kernel<true, 4, 1, false, false>": "", "conv64_wino2d_shell_kernel<4, 1, false>": "",
    "conv64_wino2d_kernel<false, 4, 2, false, true>": "fdn_conv64_fwd_mask (the forward + the sign mask of its output)", "conv64_wino2d_kernel<false, 4, 1, false, true>": "",
    "conv64_wino2d_shell_kernel<4, 2, true>": "fdn_conv64_dgrad_fused_mask (act' from the sign mask)", "conv64_wino2d_shell_kernel<4, 1, true>": "",
    "conv64_wino2d_pc_kernel<false>": "... FDN_ALGO_WINO_BF16X3: the F(4,3) x F(4,3) products as bf16 x 3 on v_mfma_f32_16x16x32_bf16 (producer / consumer waves)",
    "conv64_wino2d_pc_kernel<true>": "",
    "conv64_wino2d_kernel<false, 2, 2, false, false>": "... H even only, or FDN_ALGO_WINO_H2 (F(2,3) x F(4,3))",
    "conv64_wino2d_kernel<true, 2, 2, false, false>": "", "conv64_wino2d_shell_kernel<2, 2, false>": "",
    "conv64_wino_kernel<4, false>": "... odd H, FDN_ALGO_WINO_W (F(4,3) along W)", "conv64_wino_kernel<4, true>": "... + shell faces (FDN_DGRAD_SHELL)",
    "conv64_mfma_kernel<2, 1, 2, false>": "... W % 4 != 0, FDN_ALGO_DIRECT: the planner's three direct layouts", "conv64_mfma_kernel<2, 1, 2, true>": "",
    "conv64_mfma_kernel<1, 1, 2, false>": "", "conv64_mfma_kernel<1, 1, 2, true>": "", "conv64_mfma_kernel<1, 2, 2, false>": "", "conv64_mfma_kernel<1, 2, 2, true>": "",
    "wgrad64_wino_kernel<true>": "fdn_conv3d_wgrad 64->64, D even", "wgrad64_wino_batch_kernel<true>": "fdn_conv3d_wgrad_batch", "wgrad64_wino_kernel<false>": "... odd D, FDN_ALGO_WINO_W",
    "wgrad64_reduce_dep_kernel": "", "wgrad64_reduce_kernel": "", "wgrad64_pipe_kernel<4, 8>": "... W % 4 != 0, FDN_ALGO_DIRECT",
    "fold_halo_border_kernel": "fdn_fold_halo_border", "fold_halo_kernel": "fdn_fold_halo",
    "pack_conv64_kernel": "fdn_pack_conv64_weights", "pack_conv64_wino_kernel": "", "pack_conv64_wino2d_kernel": "", "pack_conv64_batch_kernel": "fdn_pack_conv64_weights_batch", "pack_conv64_wino44_kernel": "... the F(4,3) x F(4,3) streams, fp32 and bf16 x 3",
    "conv_cin3_fwd_mfma_kernel<T>": "fdn_conv3d_fwd 3->64", "wgrad_cin3_mfma_kernel<T>": "fdn_conv3d_wgrad 3->64", "wgrad_cin3_kernel<T>": "... odd W",
    "conv1x1_fwd_mfma_kernel<T>": "fdn_conv3d_fwd (64+64)->64 k1", "conv1x1_dgrad_mfma_kernel<T>": "fdn_conv1x1_dgrad", "wgrad_1x1_mfma_kernel<T>": "fdn_conv3d_wgrad k1",
    "head_fwd_kernel<T>": "fdn_conv3d_fwd 64->1", "head_dgrad_kernel<T>": "fdn_conv_cout1_dgrad_folded", "head_wgrad_kernel<T>": "fdn_conv3d_wgrad 64->1",
    "conv_cout1_dgrad_kernel": "fdn_conv3d_dgrad 64->1 (padded form)",
    "bias_grad_kernel<T>": "fdn_bias_grad", "reduce_partials_kernel": "", "sum_partials_kernel": "",
    "upsample_fwd_kernel<T, true>": "fdn_upsample_trilinear_fwd (input rows staged through LDS)", "upsample_fwd_kernel<T, false>": "... rows too long for the LDS", "upsample_bwd_kernel<T, 2>": "fdn_upsample_trilinear_bwd (two low-res rows per block)", "upsample_bwd_kernel<T, 1>": "... rows too long for two in the LDS", "input_features_kernel<T>": "fdn_input_features",
    "loss_main_kernel": "fdn_loss_metrics", "loss_finalize_kernel": "", "mask_sums_kernel": "", "l2_sumsq_kernel": "fdn_l2_sumsq", "l2_sumsq_partials_kernel": "fdn_l2_sumsq_partials", "adam_kernel": "fdn_adam_step",
    "gather_patches_kernel": "fdn_gather_patches",
    # conv64_bf16_kernel<MT, MODE, MULTI>, conv64_bf16_fused_kernel<MT, MULTI> (MULTI: fdn_conv64_dgrad_fused_bf16_multi)
    "conv64_bf16_kernel<8, 2, false>": "bf16 mode: fdn_conv64_fwd_bf16", "conv64_bf16_fused_kernel<8, false>": "fdn_conv64_dgrad_fused_bf16 (inner box + shell slabs, one launch)",
    "conv64_bf16_fused_kernel<4, false>": "", "conv64_bf16_kernel<8, 1, false>": "", "conv64_bf16_kernel<8, 0, false>": "",
    "conv64_bf16_kernel<4, 2, false>": "", "conv64_bf16_kernel<4, 1, false>": "", "conv64_bf16_kernel<4, 0, false>": "",
    "conv64_bf16_fused_kernel<8, true>": "fdn_conv64_dgrad_fused_bf16_multi", "conv64_bf16_fused_kernel<4, true>": "",
    "conv64_bf16_kernel<8, 2, true>": "... on grids whose fused dgrad is issued in parts", "conv64_bf16_kernel<8, 1, true>": "", "conv64_bf16_kernel<8, 0, true>": "",
    "conv64_bf16_kernel<4, 2, true>": "", "conv64_bf16_kernel<4, 1, true>": "", "conv64_bf16_kernel<4, 0, true>": "",
    "pack_conv64_bf16_kernel": "", "fold_halo_border_bf16_kernel": "",
    "wgrad64_bf16_dma_kernel": "fdn_conv3d_wgrad_bf16", "wgrad64_bf16_kernel": "... tensors of 4 GB and more",
    "wgrad64_bf16_dma_batch_kernel": "fdn_conv3d_wgrad_bf16_batch", "wgrad64_reduce_batch_kernel": "",
}


def test_product_library_holds_only_reachable_kernels(code_objects):
    names = set()
    for notes, _ in code_objects:
        for name in _kernels(notes):
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            d = re.sub(r"\(.*$", "", re.sub(r"^void ", "", d.replace("(anonymous namespace)::", "")))
            names.add(re.sub(r"<(float|unsigned short)(,|>)", r"<T\2", d))
    assert names == set(PRODUCT_KERNELS), (sorted(names - set(PRODUCT_KERNELS)), sorted(set(PRODUCT_KERNELS) - names))


def test_hot_kernels_use_the_instructions_the_design_names(code_objects):
    asm = "\n".join(a for _, a in code_objects)
    want = {"20conv64_wino2d_kernelILb0ELi4ELi2ELb0ELb0EE": "v_mfma_f32_16x16x4_f32", "20conv64_wino2d_kernelILb0ELi2ELi2ELb0ELb0EE": "v_mfma_f32_16x16x4_f32",
            "20conv64_wino2d_kernelILb0ELi4ELi2ELb0ELb1EE": "ds_bpermute_b32",
            "23conv64_wino2d_pc_kernelILb0EE": "v_mfma_f32_16x16x32_bf16", "23conv64_wino2d_pc_kernelILb1EE": "v_cvt_pk_bf16_f32",
            "19wgrad64_wino_kernelILb1": "v_mfma_f32_32x32x2_f32",
            "18conv64_bf16_kernelILi8ELi2": "v_mfma_f32_32x32x16_bf16", "23wgrad64_bf16_dma_kernel": "ds_read_b64_tr_b16"}
    for kern, ins in want.items():
        body = _function(asm, kern)
        assert body is not None, kern
        assert ins in body, (kern, ins)
    assert re.search(r"buffer_load_dwordx4 .* lds", _function(asm, "23wgrad64_bf16_dma_kernel")), "LDS-DMA staging"
    assert re.search(r"buffer_load_dwordx4 .* lds", _function(asm, "19wgrad64_wino_kernelILb1")), "LDS-DMA second plane"


def test_dma_wgrad_tile_loop_keeps_two_tiles_in_flight(code_objects):
    asm = "\n".join(a for _, a in code_objects)
    for name in ("23wgrad64_bf16_dma_kernel", "29wgrad64_bf16_dma_batch_kernel"):      # one layer per launch / several (the same body)
        body = _function(asm, name)
        lines = body.splitlines()
        first_tr = [i for i, l in enumerate(lines) if "ds_read_b64_tr_b16" in l]
        mfma = [i for i, l in enumerate(lines) if "v_mfma_f32_32x32x16_bf16" in l]
        assert len(mfma) == 36 and first_tr, name
        # between the first transposing read and the last MFMA of the loop body: counted waits only
        window = lines[first_tr[0] - 12:mfma[-1]]
        full = [l for l in window if re.search(r"s_waitcnt\s+vmcnt\(0\)", l)]
        assert not full, name + ": the compiler waits for every outstanding load inside the K loop:\n" + "\n".join(full)
        assert any(re.search(r"s_waitcnt\s+vmcnt\(6\)", l) for l in lines) and any(re.search(r"s_waitcnt\s+vmcnt\(5\)", l) for l in lines), name


def test_fp32_wgrad_tile_loops_do_not_wait_behind_their_raw_row_loads(code_objects):
    """wgrad64_wino_kernel<true>: three copies of the 54-MFMA tile loop (one per wave role).  The raw rows of tile k + 2 are requested in
    slots 6..16 and consumed a whole tile later; the only full vector-memory waits allowed are at the top of a loop (slots 0..2: the
    explicit wait in front of the scratch read-back / the first use of the rows) and at its back edge.  Round 5 saw hipcc put
    s_waitcnt vmcnt(0) behind EVERY raw-row load after an unrelated change of constants (0.60 -> 0.75 ms at (8,48^3)): this is the tripwire."""
    asm = "\n".join(a for _, a in code_objects)
    body = _function(asm, "19wgrad64_wino_kernelILb1")
    assert body is not None
    lines = body.splitlines()
    mfma = [i for i, l in enumerate(lines) if "v_mfma_f32_32x32x2_f32" in l]
    assert len(mfma) == 3 * 54, len(mfma)
    for copy in range(3):
        lo, hi = mfma[copy * 54 + 9], mfma[copy * 54 + 53]          # behind slot 2, up to the last slot
        full = [l for l in lines[lo:hi] if re.search(r"s_waitcnt\s+vmcnt\(0\)", l)]
        assert not full, "loop copy %d waits for every outstanding load inside the tile:\n%s" % (copy, "\n".join(full))
