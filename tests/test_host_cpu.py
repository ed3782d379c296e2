"""CPU-only checks of the host logic: plan construction vs the oracle's maps and the
reference goldens, and that the C-ABI library loads and exports every declared symbol."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from infinite_video_amd import _lib, basis_maps
from oracle import ltm_oracle as O
from tests.golden.cases import CASES, DENSE_CASES, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expand_first(plan):
    col = np.full(plan.T, -1, np.int32)
    for b, s, e in zip(plan.first_row_box, plan.first_row_begin, plan.first_row_end):
        col[s:e] = b
    return col


def _expand_inf(plan):
    col = np.full(plan.S + plan.T, -1, np.int32)
    for n in range(plan.N):
        for s in plan.inf_old_slot[plan.inf_old_ptr[n]:plan.inf_old_ptr[n + 1]]:
            col[s] = n
    for b, s, e in zip(plan.inf_row_box, plan.inf_row_begin, plan.inf_row_end):
        col[plan.S + s:plan.S + e] = b
    return col


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_plan_matches_reference_operator_goldens(case):
    g = load_golden(case)
    for T in sorted(set(case.chunk_T)):
        p = basis_maps.build_plan(T, case.N, case.tau)
        fc, ic = _expand_first(p), _expand_inf(p)
        np.testing.assert_array_equal(fc, g[f"T{T}_first_col"])
        np.testing.assert_array_equal(ic, g[f"T{T}_inf_col"])
        fv = np.where(fc >= 0, p.first_box_val[np.maximum(fc, 0)], 0).astype(np.float32)
        iv = np.where(ic >= 0, p.inf_box_val[np.maximum(ic, 0)], 0).astype(np.float32)
        np.testing.assert_array_equal(fv, g[f"T{T}_first_val"])
        np.testing.assert_array_equal(iv, g[f"T{T}_inf_val"])
        np.testing.assert_array_equal(p.uniform_idx, g[f"T{T}_uniform_idx"])


@pytest.mark.parametrize("T,N,tau", [(8, 64, .75), (7, 64, .75), (16, 64, .5), (256, 256, .75), (255, 256, .75),
                                     (2, 16, .9), (100, 128, .3), (32, 1024, .75)])
def test_plan_matches_oracle_maps(T, N, tau):
    p = basis_maps.build_plan(T, N, tau)
    m = O.build_maps(T, N, tau)
    np.testing.assert_array_equal(_expand_first(p), m.first_col)
    np.testing.assert_array_equal(_expand_inf(p), m.inf_col)
    np.testing.assert_array_equal(p.readout_w, m.w)
    assert p.readout_w_out == pytest.approx(m.w_out, abs=0)
    mod, edge_box, bin_box = O.sticky_bin_rows(N)
    np.testing.assert_array_equal(p.edge_box, edge_box)
    np.testing.assert_array_equal(p.bin_box, bin_box[:128])
    np.testing.assert_array_equal(p.edge_dx, (mod[1:] - mod[:-1]).astype(np.float32))
    np.testing.assert_array_equal(p.uniform_idx, m.uniform_idx)
    # every slot / frame accounted for exactly once; last frame (t = 1.0) dropped
    assert p.inf_old_ptr[-1] == len(p.inf_old_slot)
    assert _expand_inf(p)[-1] == -1


def test_plan_rejects_single_frame_chunks():
    with pytest.raises(basis_maps.UnsupportedBasis):
        basis_maps.build_plan(1, 64, .75)


def test_library_exports_every_declared_symbol():
    """include/infv_ltm.h <-> libinfv_ltm.so <-> the ctypes table, without touching a GPU."""
    header = open(os.path.join(ROOT, "include", "infv_ltm.h")).read()
    header += open(os.path.join(ROOT, "include", "infv_vqf.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(infv_(?:ltm|vqf)_[a-z_0-9]+)\s*\(", header, re.M))
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.infv_ltm_abi_version() == _lib.ABI_VERSION
    # argument validation needs no device
    assert lib.infv_ltm_create(None, None) == -1
    assert b"null" in lib.infv_ltm_last_error()
    cfg = _lib.Config(64, 12, 32, 768, 32, 1, 512, 1, 32, 8)      # head_size 32: unsupported
    h = ctypes.c_void_p()
    assert lib.infv_ltm_create(ctypes.byref(cfg), ctypes.byref(h)) == -2
    assert lib.infv_ltm_has_plan(None, 8) == -1
    assert lib.infv_vqf_create(None, None) == -1
    vcfg = _lib.VqfConfig(2, 12, 700, 3072, 768, 32, 32, 4096, 512, 0.9, 1e-12)    # hidden != 12 * 64
    assert lib.infv_vqf_create(ctypes.byref(vcfg), ctypes.byref(h)) == -2


def test_shipped_library_reads_only_the_documented_environment_options():
    """Experiment / A-B / fault-injection knobs go through ``exp_env`` (csrc/knobs.h): a constant nullptr in the shipped build, so
    their names are not even in the binary; the experiments build has them."""
    import subprocess
    def knobs(path):
        out = subprocess.run(["strings", "-a", path], capture_output=True, text=True, check=True).stdout.split("\n")
        return {w for line in out for w in re.findall(r"INFV_[A-Z0-9_]+", line)}
    documented = {"INFV_VPROJ_SPLIT", "INFV_PROJ_X6", "INFV_VQF_FP32", "INFV_VQF_FUSE", "INFV_VQF_SPLIT_CACHE_GB"}
    shipped = knobs(_lib.LIB_PATH if "exp" not in os.path.basename(_lib.LIB_PATH) else os.path.join(os.path.dirname(_lib.LIB_PATH), "libinfv_ltm.so"))
    assert shipped == documented, sorted(shipped - documented)
    exp = knobs(os.path.join(os.path.dirname(_lib.LIB_PATH), "libinfv_ltm_exp.so"))
    assert documented < exp and {"INFV_SKIP", "INFV_CHAIN_FAULT", "INFV_POOL_ROWS"} <= exp


def test_experiments_build_keeps_the_register_footprint_of_the_shipped_kernels(tmp_path):
    """Who shares a CU with whom is decided by registers per wave: an experiment branch compiled into a
    pipeline kernel must not change its allocation, or every A/B run with the experiments build measures a different pipeline
    (round 3: timing branches took the projection GEMM from 130 to 202 VGPRs and with them its seat beside a pooling
    workgroup).  Reads the VGPR counts of both libraries' gfx950 code objects."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{llvm}/llvm-objdump") and os.path.exists(f"{llvm}/llvm-readelf")):
        pytest.skip("ROCm LLVM binutils not found")
    def vgprs(lib):
        d = tmp_path / lib
        d.mkdir()
        shutil.copy(os.path.join(os.path.dirname(_lib.LIB_PATH), lib), d / lib)
        subprocess.run([f"{llvm}/llvm-objdump", "--offloading", lib], cwd=d, capture_output=True, check=True)
        out = {}
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", f], cwd=d, capture_output=True, text=True, check=True).stdout
            name = None
            for line in notes.split("\n"):
                m = re.search(r"\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.search(r"\.vgpr_count:\s+(\d+)", line)
                if m and name:
                    out[name] = int(m.group(1))
        return out
    shipped, exp = vgprs("libinfv_ltm.so"), vgprs("libinfv_ltm_exp.so")
    pipeline = [k for k in shipped if re.search(r"pool_rows2_kernel|pool_frames_kernel|gemm_nt_lw_kernel|uc_fast_kernel|"
                                                r"chain_batch3_kernel|alpha_rows2_kernelILi[12]E|build_rows_kernel", k)]   # (alpha: the table widths 4 and 8 every plan has; <0> is the fallback for wider ones)
    assert len(pipeline) >= 8
    granule = lambda v: (v + 7) // 8                                 # registers are allocated in blocks of 8
    diff = {k: (shipped[k], exp.get(k)) for k in pipeline if k not in exp or granule(exp[k]) != granule(shipped[k])}
    assert not diff, diff


def _kernel_names(lib, tmp_path):
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{llvm}/llvm-objdump") and os.path.exists(f"{llvm}/llvm-readelf")):
        pytest.skip("ROCm LLVM binutils not found")
    d = tmp_path / ("names_" + lib)
    d.mkdir()
    shutil.copy(os.path.join(os.path.dirname(_lib.LIB_PATH), lib), d / lib)
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", lib], cwd=d, capture_output=True, check=True)
    names = set()
    for f in sorted(os.listdir(d)):
        if "amdgcn" in f:
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", f], cwd=d, capture_output=True, text=True, check=True).stdout
            names |= set(re.findall(r"\.name:\s+(\S+)", notes))
    return names


def test_shipped_library_holds_only_the_kernels_it_can_launch(tmp_path):
    """A/B variants live in the experiments build only: the shipped code object has ONE persistent role-S kernel (16-row tiles,
    atomics exchange), ONE instantiation per pooling form and token type it can select, no LDS-DMA / mailbox / 8-row-tile
    variants, of round 5's call-long machinery only the one pooling launch per call and the flag_wait kernel that follows it (no
    resident role S, no resident GEMM tile queue, no flag_set / descriptor kernels), none of the kernels deleted
    in round 4 (grid-stride fused pooling, rolling double-buffered pooling, round 2-3's chain_batch2)."""
    shipped, exp = _kernel_names("libinfv_ltm.so", tmp_path), _kernel_names("libinfv_ltm_exp.so", tmp_path)
    def having(names, frag):
        return sorted(n for n in names if frag in n)
    assert len(having(shipped, "chain_batch3_kernel")) == 1 and "Lb0" in having(shipped, "chain_batch3_kernel")[0]
    assert "Lb0ELb0E" in having(shipped, "chain_batch3_kernel")[0]      # atomics exchange, register loader
    assert len(having(exp, "chain_batch3_kernel")) == 5              # + 8-row tiles, mailbox exchange, and round 6's LDS-DMA loader (128 registers: measured, not shipped)
    for gone in ("chain_batch2_kernel", "pool_rows_kernel", "pool_frames_db_kernel"):
        assert not having(shipped, gone) and not having(exp, gone), gone
    assert not having(shipped, "pool_rows2_dma_kernel") and not having(exp, "pool_rows2_dma_kernel")     # (racy LDS-DMA variant: deleted in round 5)
    assert having(shipped, "flag_wait_kernel")      # round 6: the GEMM stream holds on the call-long pooling launch's completion counts
    for exp_only in ("mailbox_to_part_kernel", "gemm_x6_call_kernel", "flag_set_kernel",
                     "chain_call_desc_kernel", "gemm_call_desc_kernel"):
        assert not having(shipped, exp_only) and having(exp, exp_only), exp_only
    assert len(having(shipped, "pool_frames_kernel")) == 4          # {padded 512-thread, plain 256-thread} x {fp32, bf16 tokens}
    assert len(having(shipped, "pool_rows2_kernel")) == 8           # {8, 4 loads per burst} x {fp32, bf16 tokens} x {rows only, rows + bf16 planes}
    # the general-psi step (Gaussian family) and the dense step are product paths
    for needed in ("psi_gemm_kernel", "psi_update_kernel", "psi_masses_kernel", "psi_grid_kernel", "psi_ctx_kernel",
                   "dense_update_kernel", "uc_fast_kernel", "alpha_rows2_kernel", "gemm_nt_lw_kernel", "gemm_x6_wide_kernel"):
        assert having(shipped, needed), needed


def test_every_kernel_launch_goes_through_the_counting_macro():
    """``infv_ltm_launch_count`` (launches per chunk in the bench line) counts what INFV_LAUNCH (csrc/knobs.h) issues: no source
    may launch a kernel any other way."""
    csrc = os.path.join(ROOT, "infinite-video_amd", "csrc")
    n_launch = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//[^\n]*", "", text)                         # (comments may mention the HIP macro)
        n_launch += code.count("INFV_LAUNCH(")
        if f == "knobs.h":
            assert code.count("<<<") == 1                             # the macro's own launch
            continue
        assert "<<<" not in code and "hipLaunchKernelGGL" not in code and "hipModuleLaunchKernel" not in code and \
               "hipExtLaunchKernel" not in code and "hipLaunchCooperativeKernel" not in code, f
    assert n_launch > 60


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "infinite-video_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_memory_file_format_rejects_foreign_files(tmp_path):
    import torch
    from safetensors.torch import save_file
    from infinite_video_amd.memory_io import load_memory
    p = str(tmp_path / "x.safetensors")
    save_file({"a": torch.zeros(2)}, p, metadata={"format": "something-else"})
    with pytest.raises(ValueError):
        load_memory(p, [], "cpu")


@pytest.mark.parametrize("case", DENSE_CASES, ids=lambda c: c.name)
def test_dense_plan_matches_reference_operators(case):
    """num_basis whose fp32 boxes (mu +/- width/2, BASIS.py:248-250) overlap at a sample position: the reference's ridge
    operator has two non-zeros in some rows.  ``build_plan`` then returns the DENSE form -- G from the reference's own ATen
    sequence, two-box tables for the resampling points -- and it must equal what the real reference built
    (tests/golden/make_goldens.py stores ``Gs[T]``, ``G_inf`` and ``samples`` of such cases as they are)."""
    g = load_golden(case)
    for T in sorted(set(case.chunk_T)):
        p = basis_maps.build_plan(T, case.N, case.tau)
        Np = basis_maps.padded_N(case.N)                          # a num_basis that is no multiple of 16 is padded with inert rows
        assert p.dense and p.first_GT.shape == (Np, T) and p.inf_GT.shape == (Np, 512 + T) and p.N_pad == (Np if Np != case.N else 0)
        assert not p.first_GT[case.N:].any() and not p.inf_GT[case.N:].any() and not p.readout_w[case.N:].any()
        # (Tensor.inverse is LAPACK: allow the last bits to depend on the host's BLAS kernels)
        np.testing.assert_allclose(p.first_GT[:case.N].T, g[f"T{T}_first_G"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(p.inf_GT[:case.N].T, g[f"T{T}_inf_G"], rtol=0, atol=1e-6)
        if case.N % 16 == 0:
            assert int((p.inf_GT != 0).sum(0).max()) == 2        # the reason the sparse form does not apply
        assert p.psi == (case.N == 37)                            # a read-out grid point in two boxes: general-psi step
        smp = g[f"T{T}_uniform_samples"]                          # [S, N] psi of the uniform resampling positions
        for s_, (a, b) in enumerate(p.uniform_box2):
            assert sorted(np.flatnonzero(smp[s_]).tolist()) == sorted(x for x in (int(a), int(b)) if x >= 0)


def test_gaussian_plan_operators_match_the_reference_chain():
    """basis_maps.build_gaussian_plan: both dense ridge operators of every chunk length of the Gaussian-family chains equal what
    the REAL reference built (golden gauss_chain.npz / gauss_uniform.npz), and psi is a dense, strictly positive row wherever
    the step evaluates it."""
    from tests.golden.cases import GAUSS_CASES, GAUSS_SIGMAS, load_golden
    for case in GAUSS_CASES:
        g = load_golden(case)
        for T in sorted(set(case.chunk_T)):
            p = basis_maps.build_gaussian_plan(T, case.N, case.tau, tuple(GAUSS_SIGMAS))
            assert p.dense and p.psi
            # (the N x N inverse comes from LAPACK on the host, as in the reference: same library here, so the bits agree)
            np.testing.assert_allclose(p.first_GT.T, g[f"T{T}_first_G"], rtol=0, atol=1e-6)
            np.testing.assert_allclose(p.inf_GT.T, g[f"T{T}_inf_G"], rtol=0, atol=1e-6)
            assert p.psi_edge.shape == (129, case.N) and p.psi_bin.shape == (128, case.N)
            assert p.psi_uniform.shape == (512, case.N) and p.psi_grid.shape == (1000, case.N)
            assert float(p.psi_grid.max()) > 1.0 and float(p.psi_grid.min()) >= 0.0
            assert abs(float(p.grid_w.sum()) - 1.0) < 1e-6


def test_gaussian_operator_matches_the_reference_family():
    """Second basis family at operator level: ``compute_G`` of the reference with its own ``GaussianBasisFunctions`` (the golden
    comes from the real module with its builder hook pointed at ``add_gaussian_basis_functions``,
    tests/golden/make_gaussian_goldens.py) -- a fully dense ridge operator."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "gauss_operator.npz"))
    GT = basis_maps.gaussian_first_operator_T(int(g["T"]), int(g["N"]), [float(x) for x in g["sigmas"]])
    assert GT.shape == (int(g["N"]), int(g["T"]))
    assert float((g["G_first"] != 0).mean()) > 0.9                       # dense, unlike the box operators
    np.testing.assert_allclose(GT.T, g["G_first"], rtol=0, atol=1e-4 * float(np.abs(g["G_first"]).max()))   # LAPACK-dependent


def test_every_multiple_of_16_has_a_plan_and_sparse_ones_stay_sparse():
    """``UnsupportedBasis`` is gone for the multiples of 16 the kernels accept (N <= 256); values whose boxes partition
    the samples keep the sparse closed form."""
    for N in range(16, 257, 16):
        for T in (8, 255, 256):
            p = basis_maps.build_plan(T, N, .75)
            assert p.N == N
    for N in (64, 128, 144, 256):
        assert not basis_maps.build_plan(256, N, .75).dense
    for N in (48, 96, 192):
        assert basis_maps.build_plan(256, N, .75).dense
    # a read-out grid point in two boxes (432, 37, ...): no closed-form read-out -- the plan carries the rectangular psi itself
    # as dense 0/1 rows and the step takes the general-psi form (432 itself is beyond the kernels' N <= 256)
    p = basis_maps.build_plan(16, 432, .75)
    assert p.dense and p.psi and p.psi_grid.shape == (1000, 432) and float(p.psi_grid.sum(1).max()) == 2.0


def test_any_num_basis_up_to_256_has_a_plan():
    """The reference takes any ``--num_basis`` (run_inference_inf_video_llama_nextqa.py:61).  Values that are no multiple of the
    kernels' 16-wide tile get the dense form padded with inert basis functions (zero operator rows, zero read-out weight);
    the real ones are untouched: the tables never refer to a padding index."""
    for N in (2, 7, 20, 37, 50, 100, 129, 250, 255):
        for T in (6, 8, 16):
            p = basis_maps.build_plan(T, N, .75)
            Np = basis_maps.padded_N(N)
            assert p.dense and p.N == N and p.N_pad == Np and Np % 16 == 0 and Np - N < 16
            assert p.first_GT.shape == (Np, T) and not p.first_GT[N:].any() and not p.inf_GT[N:].any()
            for tab in (p.bin_box2, p.edge_box2, p.uniform_box2):
                assert int(tab.max()) < N
            assert p.readout_w.shape == (Np,) and not p.readout_w[N:].any()
            if p.psi:
                assert p.psi_grid.shape == (1000, Np) and not p.psi_grid[:, N:].any() and not p.psi_edge[:, N:].any()


def test_bench_gpus_flag_spawns_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` with no torchrun environment must start N ranks as a child job
    (torch.distributed.run, 127.0.0.1 rendezvous) and leave the GPU to them."""
    import subprocess
    import bench
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_short_memory_buffer_keeps_the_reference_frames():
    """Producer layout (infinityqa.py:251-278,285-307): frames with cur_frame <= n_frame are kept, encode_video then
    drops the oldest beyond n_position**2, n_position = min(32, ceil(sqrt(T)))."""
    import math
    import torch
    from infinite_video_amd.video_qformer import ShortMemoryBuffer
    P, d = 4, 8
    frames = torch.arange(1100 * P * d, dtype=torch.float32).reshape(1100, P, d)
    buf = ShortMemoryBuffer(P, d, capacity_frames=1101)
    for F, n_frame in ((5, 2048), (8, 2048), (300, 256), (1100, 2048), (10, 3)):
        buf.replace(frames[:F], n_frame)
        # the reference: a list of the frames with index <= n_frame ...
        ref = [frames[i] for i in range(F) if i <= n_frame]
        assert len(buf) == len(ref)
        # ... of which encode_video keeps the newest n_position^2
        n_position = min(32, math.ceil(math.sqrt(len(ref))))
        while len(ref) > n_position * n_position:
            ref.pop(0)
        want = torch.cat([f.unsqueeze(0) for f in ref], 0).reshape(1, -1, d)
        got = buf.frames()
        assert got.data_ptr() >= buf.store.data_ptr() and torch.equal(got, want)          # a view, same numbers
        assert ShortMemoryBuffer.frame_cap(len(buf)) == (n_position, len(ref))
    half = ShortMemoryBuffer(P, d, capacity_frames=16, dtype=torch.bfloat16).replace(frames[:9] / 1000.0)
    assert half.frames().dtype == torch.bfloat16 and half.frames().shape == (1, 9 * P, d)
    with pytest.raises(ValueError):
        buf.replace(frames[:3, :2])
    with pytest.raises(RuntimeError):
        ShortMemoryBuffer(P, d, 4).frames()


def test_bench_reads_the_honest_ceiling_from_the_committed_pmc_pass():
    """bench.py's ``roofline.fabric_bytes_per_chunk``: bytes that crossed the fabric per chunk for every kernel of the whole-video
    pipeline, from the newest committed PMC summary (no GPU involved).  The pooling stream's share must equal its algorithmic bytes
    (no wasted re-reads), the total must sit between the pooling's bytes and the section-8d formula's 39.7 MB, and the launch
    census the drop-in / encode_video legs use must be exported."""
    import bench
    fab = bench.pmc_fabric_bytes_per_chunk(42)
    assert fab is not None and fab["source"].endswith("_pmc_summary.json")
    pool = fab["per_kernel_high"]["pool_rows2_kernel"]
    assert abs(pool - bench.BYTES_POOL_PER_CHUNK) / bench.BYTES_POOL_PER_CHUNK < 0.01, (pool, bench.BYTES_POOL_PER_CHUNK)
    assert bench.BYTES_POOL_PER_CHUNK < fab["low"] <= fab["high"] < bench.BYTES_PER_CHUNK
    for k in ("gemm_x6_wide_kernel", "uc_fast_kernel", "alpha_rows2_kernel", "chain_batch3_kernel"):
        assert fab["per_kernel_high"][k] > 0, k
    assert "split3_rows_kernel" not in fab["per_kernel_high"]          # (once per call since round 5: the weights' planes)
    # the pooling kernel's largest launch of the committed summary and the chunks it covered (round 6: one launch per long call)
    traffic, src, chunks = bench.pmc_traffic_per_full_launch()
    assert src.endswith("_pmc_summary.json") and chunks and chunks >= 42
    assert abs(traffic / chunks - bench.BYTES_POOL_PER_CHUNK) / bench.BYTES_POOL_PER_CHUNK < 0.01
    lib = _lib.load()
    assert int(lib.infv_ltm_launch_count()) >= 0
