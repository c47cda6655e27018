"""The tile planner behind e4t_gemm_plan / e4t_conv3x3_plan / e4t_gemm_tn_plan (csrc/gemm.hip: plan_gemm) on the training step's own
shapes.  Pure host logic of the library (no launch, no GPU: the CU count defaults to the MI355X's 256), so the rules DESIGN.md §2.1
states — which kernel family takes which shape, when split-K is used, how much workspace a caller must bring — are pinned on CPU.
The timings that justify the rules are GPU measurements (profiles/r03_sweep_tiles_c.txt); this test keeps the rules from drifting."""
import ctypes as C
import os

import pytest

from e4t import _C

lib = _C.load()


def gemm_plan(M, N, K, flags=0, tile=0, splitk=0, batch=1, rowbias=False, rows_per_batch=0, panels=(0, 0, 0)):
    d = _C.GemmDesc(M=M, N=N, K=K, K1=K, lda=K, ldb=K, ldc=N, batch=batch, alpha=1.0, flags=flags, tile=tile, splitk=splitk,
                    rows_per_batch=rows_per_batch, rowbias=(1 << 20) if rowbias else None,      # planner only tests pointers for NULL
                    panel_rows=panels[0], panel_stride=panels[1], panel_off=panels[2])
    pl = _C.GemmPlan()
    assert lib.e4t_gemm_plan(C.byref(d), C.byref(pl)) == 0, lib.e4t_last_error()
    return pl


def conv_plan(B, H, W, Cin, Cout, mode=_C.CONV_S1, Hout=None, Wout=None, tile=0, splitk=0):
    d = _C.ConvDesc(B=B, Hin=H, Win=W, Cin=Cin, Hout=Hout or H, Wout=Wout or W, Cout=Cout, mode=mode, tile=tile, splitk=splitk)
    pl = _C.GemmPlan()
    assert lib.e4t_conv3x3_plan(C.byref(d), C.byref(pl)) == 0, lib.e4t_last_error()
    return pl


@pytest.mark.parametrize("shape, tile, dims", [
    # SD-1.4, B = 16 (profiles/r03_roofline_per_shape.csv)
    ((65536, 320, 2560), 2320, (256, 320)),      # ff.net.2 at the 64x64 level: 256 x 320 ping-pong, full rounds
    ((65536, 320, 1280), 2320, (256, 320)),
    ((4096, 10240, 1280), 2320, (256, 320)),     # GEGLU projection at the 16x16 level
    ((16384, 5120, 640), 2320, (256, 320)),
    ((65536, 2560, 320), 5256, (256, 128)),      # K = 320: too short for the ping-pong prologue, N % 128 == 0
    ((4096, 1280, 1280), 160, (128, 160)),       # 256 workgroups of 128 x 160: one per CU, three LDS stages (23 vs 27 us on 128 x 128)
    ((16384, 640, 640), 160, (128, 160)),
    ((4112, 3840, 1280), 512, (256, 256)),       # ViT qkv
])
def test_step_gemms_get_the_documented_tile(shape, tile, dims):
    pl = gemm_plan(*shape)
    assert (pl.tile, pl.tile_m, pl.tile_n) == (tile, *dims), (shape, pl.tile, pl.tile_m, pl.tile_n)
    assert pl.splitk == 1 and pl.workspace_bytes == 0


@pytest.mark.parametrize("shape, tile, splitk", [
    # grids under a few rounds (the B = 1 / B = 4 steps of BASELINE configs[4], the text encoder, the 8 x 8 level): tile, LDS depth and
    # split-K come from the launch cost model (gemm.hip: small_grid_plan; measured in tools/sweep_small_m.py)
    ((576, 1280, 1280), 64, 1),                  # 180 workgroups of 64 x 64, deep pipeline: 8.8 us (round 3: 13.9)
    ((257, 3840, 1280), 64, 1),                  # ViT qkv at B = 1
    ((1232, 768, 3072), 64, 1),                  # text encoder fc2: 240 workgroups, 48 K-tiles, no split (18 vs 22 us with 3 splits)
    ((1232, 3072, 768), 128, 1),                 # fc1: 960 workgroups of 64 x 64 would need two rounds; 240 of 128 x 128 with four stages
    ((1024, 1280, 5120), 128, 3),                # 8 x 8 level, K-deep: 80 tiles x 3 splits
    ((16384, 640, 640), 160, 1),
    ((1024, 10240, 1280), 160, 1),               # 512 workgroups of 128 x 160 = one round of two per CU (36 vs 52 us on 128 x 128)
])
def test_small_grids_follow_the_cost_model(shape, tile, splitk):
    pl = gemm_plan(*shape)
    assert (pl.tile, pl.splitk) == (tile, splitk), (shape, pl.tile, pl.splitk)
    assert pl.workspace_bytes == (splitk * shape[0] * shape[1] * 4 if splitk > 1 else 0)


def test_cost_model_respects_what_each_instantiation_can_do():
    # the GENERAL epilogue (exact GELU, per-row row-bias lookup) exists for the 64 x 64 (3-stage) and 128 x 128 (2-stage) tiles only
    for shape in [(257, 5120, 1280), (576, 5120, 1280), (1028, 5120, 1280), (77, 3072, 768)]:
        assert gemm_plan(*shape, flags=_C.ACT_GELU).tile in (64, 128), shape
        assert gemm_plan(*shape, rowbias=True, rows_per_batch=77).tile in (64, 128), shape
    # N % 160 != 0 never gets the 128 x 160 tile; ragged M / N / K are priced by whole tiles
    for shape in [(4096, 1024, 1280), (1000, 200, 328), (130, 72, 1032), (4097, 1281, 1283)]:
        pl = gemm_plan(*shape)
        assert pl.tile in (64, 128) and pl.tile_n == pl.tile, (shape, pl.tile)
    # an explicit split is kept as asked (only the tile and its depth are chosen around it); an explicit tile bypasses the model
    pl = gemm_plan(1024, 1280, 5120, splitk=2)
    assert pl.splitk == 2 and pl.workspace_bytes == 2 * 1024 * 1280 * 4
    assert gemm_plan(576, 1280, 1280, tile=128).tile == 128 and gemm_plan(4096, 1280, 1280, tile=3128).tile == 128
    # split-K keeps at least four K-tiles per split and never splits a one-K-tile GEMM
    assert gemm_plan(77, 1024, 64).splitk == 1 and gemm_plan(16, 1280, 192).splitk == 1
    # batched and batch-reducing GEMMs (the grouped E4T head) stay with the rules: the model prices single GEMMs
    assert gemm_plan(1280, 1280, 8, batch=129, flags=_C.OUT_F32 | _C.REDUCE_BATCH).splitk >= 1


def test_deep_k_small_grid_uses_ping_pong_with_split_k_and_reports_the_workspace():
    M, N, K = 4096, 1280, 10240                  # 80 tiles of 256 x 256 on 256 CUs, 160 K-tiles
    pl = gemm_plan(M, N, K)
    assert pl.tile == 512 and pl.splitk == 3
    assert pl.workspace_bytes == pl.splitk * M * N * 4       # fp32 partials, one per split


@pytest.mark.parametrize("args, tile", [
    ((16, 512, 512, 128, 128), 5256),            # VAE 128-channel convs: N = 128
    ((16, 256, 256, 256, 256), 512),             # VAE / UNet convs with N % 256 == 0 and many tiles: 256 x 256 ping-pong
    ((16, 128, 128, 512, 512), 512),
    ((16, 64, 64, 320, 320), 2320),              # every 320-multiple width with full rounds
    ((16, 64, 64, 640, 320), 2320),
    ((16, 32, 32, 640, 640), 160),               # 64 x 2 tiles of 256 x 320 would leave half the chip idle
])
def test_step_convs_get_the_documented_tile(args, tile):
    assert conv_plan(*args).tile == tile, (args, conv_plan(*args).tile)


def test_small_map_convs_split_k():
    pl = conv_plan(16, 16, 16, 1280, 1280)       # 16 x 16 level: 80 ping-pong tiles, K = 11520
    assert pl.tile == 512 and pl.splitk == 3 and pl.workspace_bytes == 3 * 4096 * 1280 * 4
    pl = conv_plan(16, 8, 8, 1280, 1280)         # 8 x 8 level: 80 tiles of 128 x 128 with four LDS stages, split 3 ways (240 workgroups, one per CU)
    assert pl.tile == 128 and pl.splitk == 3


def test_general_epilogue_tiles():
    # exact GELU and the per-row row-bias lookup (GENERAL epilogue, gemm_common.h) exist as 64 / 128 / 160 / 256 x 256 / 256 x 320 (GEMM only)
    # instantiations.  The ViT's fc1 at M = 16 x 257 rows is 17 panels of 256: a second round for 16 rows — since round 5 the planner
    # hands those 16 rows to the tail stage and runs the other 4096 as ONE round of 256 x 320 tiles (test_tail_rows below) ...
    pl = gemm_plan(4112, 5120, 1280, flags=_C.ACT_GELU)
    assert (pl.tile, pl.tail_rows) == (2320, 16)
    # ... as it does for 16 full panels (the patch tokens; e4t_gemm_desc.panel_*)
    pl = gemm_plan(4096, 5120, 1280, flags=_C.ACT_GELU, panels=(256, 257, 1))
    assert (pl.tile, pl.splitk, pl.workspace_bytes) == (2320, 1, 0)
    assert gemm_plan(65536, 320, 2560, flags=_C.ACT_GELU).tile == 2320
    assert gemm_plan(65536, 320, 2560, rowbias=True, rows_per_batch=4097).tile == 2320      # rows_per_batch % 32 != 0
    assert gemm_plan(65536, 320, 2560, rowbias=True, rows_per_batch=4096).tile == 2320
    assert conv_plan(16, 64, 64, 320, 320).tile == 2320


def test_a_forced_tile_and_split_are_honoured_or_replaced_by_one_that_fits():
    assert gemm_plan(65536, 320, 2560, tile=128).tile == 128
    assert gemm_plan(65536, 320, 2560, tile=160).tile == 160
    pl = gemm_plan(4096, 1280, 10240, tile=128, splitk=4)
    assert pl.tile == 128 and pl.splitk == 4 and pl.workspace_bytes == 4 * 4096 * 1280 * 4
    assert gemm_plan(4096, 1000, 1280, tile=2320).tile != 2320        # N % 320 != 0: the code does not fit the shape


def test_tn_plan_reports_split_k_over_the_rows():
    d = _C.GemmDesc(M=320, N=320, K=65536, K1=65536, lda=320, ldb=320, ldc=320, batch=1, alpha=1.0, flags=_C.OUT_F32)
    pl = _C.GemmPlan()
    assert lib.e4t_gemm_tn_plan(C.byref(d), C.byref(pl)) == 0, lib.e4t_last_error()
    assert pl.splitk > 1 and pl.workspace_bytes == pl.splitk * 320 * 320 * 4


def test_tail_rows():
    """M = 128 k + r, r <= 32 (the CLIP-ViT's 16 x 257 = 4112 token rows): the plan is the plan of the first M - r rows, the r tail rows
    are computed at the end of the same launch (gemm_common.h: gemm_tail) — every tiling of 4112 rows paid a whole extra round of
    workgroups for the last 16 (same N / K at M = 4096 vs 4112 in the round-4 step: 24.7 vs 44.9 us, 73.6 vs 125.5 us, 58.7 vs 94.6 us)."""
    for (M, N, K, flags), (tile, tail) in {
            (4112, 1280, 1280, _C.OUT_F32 | _C.RES_F32): (160, 16),      # ViT attention out-proj (fp32 residual stream)
            (4112, 1280, 5120, _C.OUT_F32 | _C.RES_F32): (160, 16),      # ViT fc2: no split-K (and no 84 MB of fp32 partials) any more
            (4112, 5120, 1280, _C.ACT_GELU): (2320, 16),                 # ViT fc1: one round of 256 x 320 tiles
            (4112, 3840, 1280, 0): (512, 16),                            # ViT qkv: 240 ping-pong tiles + tail (was 255 tiles)
            (8224, 1280, 5120, 0): (160, 32),                            # B = 32: 64 x 128 + 32
    }.items():
        pl = gemm_plan(M, N, K, flags=flags)
        ref = gemm_plan(M - tail, N, K, flags=flags)
        assert (pl.tile, pl.tail_rows, pl.splitk, pl.workspace_bytes) == (tile, tail, 1, 0), (M, N, K, pl.tile, pl.tail_rows, pl.splitk)
        assert (ref.tile, ref.tile_m, ref.tile_n, ref.tail_rows) == (pl.tile, pl.tile_m, pl.tile_n, 0)
    # no tail: remainder too large / M too small / the plan of the main rows is not a single pass of a kernel that carries the tail code
    assert gemm_plan(1232, 768, 3072).tail_rows == 0                    # text encoder: 1232 = 9 x 128 + 80
    assert gemm_plan(257, 3840, 1280).tail_rows == 0                    # ViT at B = 1
    assert gemm_plan(4112, 1000, 1280).tail_rows == 0                   # N fits none of the 160 / 256 / 320-wide tiles
    assert gemm_plan(2056, 1280, 5120).tail_rows == 0                   # B = 8: the 2048 main rows want split-K on the 128 x 128 tile
    assert gemm_plan(4112, 1280, 1280, splitk=3).tail_rows == 0         # an explicit split
    assert gemm_plan(4112, 1280, 1280, rowbias=True, rows_per_batch=257).tail_rows in (0, 16)     # (allowed: the scalar epilogue handles row biases)
    assert gemm_plan(4096 + 32, 1280, 1280).tail_rows == 32 and gemm_plan(4096 + 33, 1280, 1280).tail_rows == 0
