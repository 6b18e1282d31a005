"""CPU-side checks of the product's host logic: the C-ABI library loads and exports every symbol the header declares,
the plugin registry / option surface mirror the reference, sharding + single-collective gather work under gloo, and
the product refuses to run without a GPU (no fallback)."""
import argparse
import os
import re
import subprocess
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from latentaugment_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from latentaugment_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'latentaug_hip.h')).read()
    declared = set(re.findall(r'\b(la_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/latentaug_hip.h but not exported'
        assert name in _lib.SIGNATURES, f'{name} has no ctypes signature'
    assert lib.la_abi_version() == 1


def test_product_library_has_no_development_switches(lib):
    """Kernel-variant knobs and LA_* environment switches exist in the development build (-DLA_DEV) only: the product library neither
    exports la_dev_knob_set nor contains the names of the environment variables the development build reads."""
    from latentaugment_amd import _lib
    assert not hasattr(lib, 'la_dev_knob_set')
    blob = open(_lib.LIB_PATH, 'rb').read()
    for name in (b'LA_DEV_KNOBS', b'LA_HALO_W3', b'LA_FLAT_W3', b'LA_FORCE_MT', b'LA_NO_XS_HANDOFF', b'LA_NO_SEAM_FUSE', b'LA_NO_RGB_FUSE',
                 b'LA_NO_ZT_PITCH'):
        assert name not in blob, name
    src = open(os.path.join(ROOT, 'latentaugment_amd', '_lib.py')).read()
    assert 'os.environ' not in src      # the loader takes no path from the environment


def test_no_packed_fp32_arithmetic_in_any_kernel(lib, tmp_path):
    """Round 4 traced the 'two streams disturb each other' non-reproducibility of rounds 2-3 to packed-FP32 arithmetic (v_pk_fma_f32 /
    v_pk_mul_f32 / v_pk_add_f32, emitted by the SLP vectoriser and by vector-typed float expressions): while kernels of another
    hardware queue run on the chip, the high half of such a result comes out wrong for 16-lane groups.  The library is
    therefore built without them: every gfx950 code object embedded in the product .so is disassembled here and must contain none."""
    import subprocess
    from latentaugment_amd import _lib
    bundler, objdump = '/opt/rocm/lib/llvm/bin/clang-offload-bundler', '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not (os.path.isfile(bundler) and os.path.isfile(objdump)):
        pytest.skip('LLVM tools of the ROCm image not found')
    blob = open(_lib.LIB_PATH, 'rb').read()
    starts = [m.start() for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', blob)]
    assert len(starts) >= 8
    kernels = mfma = 0
    for i, p in enumerate(starts):
        src, obj = tmp_path / f'b{i}.bin', tmp_path / f'c{i}.o'
        src.write_bytes(blob[p:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        r = subprocess.run([bundler, '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--input={src}', f'--output={obj}', '--unbundle'],
                           capture_output=True)
        assert r.returncode == 0 and obj.stat().st_size > 0, r.stderr.decode()[-500:]
        dis = subprocess.run([objdump, '-d', str(obj)], capture_output=True).stdout.decode(errors='replace')
        bad = re.findall(r'v_pk_(?:fma|mul|add)_f32', dis)
        assert not bad, f'code object {i}: {len(bad)} packed-FP32 instructions'
        kernels += len(re.findall(r's_endpgm', dis)); mfma += len(re.findall(r'v_mfma_', dis))
    assert kernels > 100 and mfma > 500      # (the disassembly really covered the kernels)


def test_pure_host_entry_points(lib):
    assert lib.la_synth_num_ws(256) == 14 and lib.la_synth_num_ws(512) == 16 and lib.la_synth_num_ws(1024) == 18
    assert lib.la_synth_num_params(4) == 10 and lib.la_synth_num_params(256) == 94
    # (in*up + pad0 + pad1 - taps + down) // down   (upfirdn2d.cpp:35-36)
    assert lib.la_upfirdn2d_out_size(128, 2, 1, 2, 1, 4) == 256
    assert lib.la_upfirdn2d_out_size(257, 1, 1, 1, 1, 4) == 256
    assert lib.la_upfirdn2d_out_size(256, 1, 2, 1, 1, 4) == 128
    assert lib.la_modconv_ds_tiles(256) == 512 and lib.la_modconv_ds_tiles(4) == 1
    import ctypes as C
    ch = (C.c_int * 7)(512, 512, 512, 512, 512, 256, 128)
    nbytes = lib.la_synth_workspace_bytes(256, 2, 512, ch, 8)
    assert 1 << 30 < nbytes < 8 << 30      # config-f 256^2, B=8: a few GB of activations + packed weights


def test_no_cpu_fallback():
    from latentaugment_amd import _lib, ops
    from latentaugment_amd.latent_aug import LatentAug
    with pytest.raises(_lib.LatentAugHipError):
        ops.bias_act(torch.zeros([2, 3]), None)
    with pytest.raises(_lib.LatentAugHipError):
        ops.upfirdn2d(torch.zeros([1, 1, 8, 8]), ops.setup_filter([1, 3, 3, 1]))
    opt = types.SimpleNamespace(img_resolution=32, batch_size=2, modalities_aug='A,B', opt_num_epochs=1, opt_lr=0.01,
                                truncation_psi=1.0, w_pix=0.0, w_lpips=0.0, w_latent=0.0, w_disc=0.0, crop_size_aug=8,
                                preprocess_aug='center_random_crop', soft_aug=False, alpha=1.0, verbose_log=False)
    with pytest.raises(_lib.LatentAugHipError):
        LatentAug('train', opt, '/tmp', [], generator={})


def test_product_never_imports_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, 'latentaugment_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(d, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M):
                    bad.append(f)
    assert not bad, bad


def test_registry_and_options_match_reference_surface():
    from latentaugment_amd.augments import find_augment_using_name, get_option_setter
    from latentaugment_amd.augments.base_aug import BaseAugment
    cls = find_augment_using_name('latent')
    assert cls.__name__ == 'LatentAugment' and issubclass(cls, BaseAugment)
    for m in ('set_input', 'forward', 'get_output', 'get_latent_input', 'get_latent_output', 'sanity_check',
              'sample_from_inversion', 'sample_from_randn', 'modify_commandline_options'):
        assert hasattr(cls, m), m
    p = get_option_setter('latent')(argparse.ArgumentParser(), True)
    opt = p.parse_args(['--model_dir', 'm', '--interim_dir', 'i'])
    # names + defaults at augments/latent_aug.py:57-96 of the reference
    want = dict(gpu_ids_aug='0', img_resolution=256, truncation_psi=1.0, rand_aug=False, lower_bound_clip=False, step_img=20,
                step_w=5, lpips_script='lpips_script', opt_num_epochs=10, opt_lr=0.01, init_w='random', crop_size_aug=64,
                preprocess_aug='center_random_crop', w_pix=1.0, w_lpips=1.0, w_latent=1.0, w_disc=1.0, p_thres=1.0,
                soft_aug=False, alpha=1.0, verbose_log=False, modalities_aug='MR_nonrigid_CT,MR_MR_T2')
    for k, v in want.items():
        assert getattr(opt, k) == v, k
    for k in ('dataset_aug', 'dataset_name_aug', 'exp_stylegan', 'network_pkl_stylegan', 'dataset_w_name', 'exp_inv',
              'network_pkl_inv'):
        assert hasattr(opt, k)
    with pytest.raises(ImportError):
        find_augment_using_name('nonexistent')


def test_crop_geometry_and_shards():
    from latentaugment_amd.latent_aug import center_crop_geometry, shard_bounds
    assert center_crop_geometry(256) == (181, 38)      # round(37.5) -> 38 (python banker's rounding, as torchvision)
    assert center_crop_geometry(512) == (362, 75)
    assert center_crop_geometry(1024) == (724, 150)
    assert [shard_bounds(8, 2, r)[:2] for r in range(2)] == [(0, 4), (4, 8)]
    assert [shard_bounds(5, 4, r)[:2] for r in range(4)] == [(0, 2), (2, 4), (4, 5), (5, 5)]
    assert [shard_bounds(32, 8, r)[:2] for r in range(8)][-1] == (28, 32)


def test_synthetic_state_dict_matches_reference_names():
    from latentaugment_amd.synthetic import make_generator_state_dict
    sd, meta = make_generator_state_dict(img_resolution=16, img_channels=2, channel_base=256, channel_max=16, w_dim=32)
    from oracle import sg2_networks as nets
    G = nets.make_generator(img_resolution=16, img_channels=2, channel_base=256, channel_max=16, w_dim=32)
    missing, unexpected = G.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all('resample_filter' in k for k in missing), missing
    assert meta['num_ws'] == G.num_ws == 6


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import random
from latentaugment_amd.latent_aug import shard_bounds, gather_shards, broadcast_controls, get_params
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
for B in (5, 8, 1):
    full = torch.arange(B * 6, dtype=torch.float32).reshape(B, 6)
    lo, hi, per = shard_bounds(B, 2, rank)
    local = full[lo:hi] * 2.0          # stand-in for the per-rank optimisation of its own samples
    out = gather_shards(local, per, B)
    assert torch.equal(out, full * 2.0), (B, rank, out)
# per-forward host draws: every rank ends with RANK 0's crop position and noise seed, whatever its own RNG state
random.seed(100 + rank)
mine = get_params(256, 64)['crop_pos']
cx, cy, seed = broadcast_controls(mine)
random.seed(100)
want = get_params(256, 64)['crop_pos']
assert (cx, cy) == tuple(want) and seed == random.getrandbits(48), (rank, cx, cy, seed)
dist.barrier()
dist.destroy_process_group()
print('ok', rank)
'''


def test_shard_gather_gloo_world2(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert 'ok' in o


def test_torchscript_vgg16_mapping(tmp_path):
    """A local TorchScript `vgg16.pt` (the reference torch.jit.load()s NVIDIA's from a URL, util_latent_aug.py:35-43, and calls it
    with resize_images=False, return_lpips=True, :395): the loader recognises the 13 convolutions, the five LPIPS channel weights
    and the input layer from the tensors of a synthetic scripted module, and exactly ONE of its candidates reproduces the module's
    own output -- which one depends on whether the script stores the lin weights or their square roots, so nothing is assumed."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers_script import eval_ops_cpu, save_scripted_vgg
    from latentaugment_amd import _lib
    from latentaugment_amd.synthesis import vgg16_from_torchscript
    for lin_sqrt in (False, True):
        path = tmp_path / f'vgg16_{int(lin_sqrt)}.pt'
        save_scripted_vgg(path, width=8, seed=3, lin_sqrt=lin_sqrt)
        cands = vgg16_from_torchscript(str(path))
        assert [op[0] for op in cands[0].ops].count('conv') == 13 and [op[0] for op in cands[0].ops].count('tap') == 5
        assert [op[0] for op in cands[0].ops].count('maxpool') == 4
        assert abs(cands[0].pre_scale[0] - 1 / 58.395) < 1e-7 and abs(cands[0].pre_shift[2] + 103.53 / 57.375) < 1e-5
        x = torch.rand([2, 1, 32, 32], generator=torch.Generator().manual_seed(1)).repeat(1, 3, 1, 1) * 255
        want = torch.jit.load(str(path))(x, resize_images=False, return_lpips=True)
        good = []
        for c in cands:
            xx = x * torch.tensor(c.pre_scale).reshape(1, 3, 1, 1) + torch.tensor(c.pre_shift).reshape(1, 3, 1, 1)
            good.append(float((eval_ops_cpu(c.ops, xx) - want).norm() / want.norm()) < 1e-5)
        assert good.count(True) == 1 and cands[good.index(True)].lin_is_sqrt == lin_sqrt
    # a scripted module that is not a VGG16 is refused
    m = torch.jit.script(torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1)))
    with pytest.raises(_lib.LatentAugHipError):
        vgg16_from_torchscript(m)


def test_curve_png_writer(tmp_path):
    """The verbose_log curves (reference snapshot_stats, util_latent_aug.py:620-633) without matplotlib: a valid greyscale PNG whose
    polyline rises where the values rise."""
    import struct
    import zlib
    import numpy as np
    from latentaugment_amd.latent_aug import write_curve_png
    path = tmp_path / 'losses_loss.png'
    write_curve_png(str(path), [0.0, 1.0, 4.0, 9.0])
    blob = open(path, 'rb').read()
    assert blob[:8] == b'\x89PNG\r\n\x1a\n'
    w, h = struct.unpack('>II', blob[16:24])
    n = struct.unpack('>I', blob[33:37])[0]
    img = np.frombuffer(zlib.decompress(blob[41:41 + n]), np.uint8).reshape(h, w + 1)[:, 1:]
    dark = np.argwhere(img[30:-30, 30:-30] == 0)                      # the markers (frame excluded)
    first, last = dark[dark[:, 1] < 40], dark[dark[:, 1] > dark[:, 1].max() - 10]
    assert first[:, 0].mean() > last[:, 0].mean() + 100           # image rows grow downwards: the last value sits far above the first
    write_curve_png(str(path), [3.0])                              # a single epoch still gives a picture
    write_curve_png(str(path), [])


def test_stream_lane_rule():
    """LatentAug.lanes_eligible (host logic, no GPU): two lanes take the FULL, even local batch of >= 4; with the discriminator a multiple
    of 8 (MinibatchStd groups sample n with n + b/4, n + 2b/4, n + 3b/4: the even / odd halves keep them only then); 'auto' not with the
    perceptual criterion (the discriminator and the perceptual branch run side by side inside one loop instead); 1 never."""
    import types
    from latentaugment_amd.latent_aug import LatentAug

    def rule(b, max_local, lanes='auto', w_disc=0.0, w_lpips=0.0):
        me = types.SimpleNamespace(stream_lanes=lanes, _max_local=max_local, w_disc=w_disc, w_lpips=w_lpips)
        return LatentAug.lanes_eligible(me, b)

    assert rule(8, 8) and rule(4, 4) and rule(16, 16) and rule(6, 6)
    assert not rule(2, 2) and not rule(5, 5) and not rule(4, 8)            # too small, odd, not the full batch
    assert rule(8, 8, w_disc=0.01) and rule(16, 16, w_disc=0.01) and not rule(4, 4, w_disc=0.01) and not rule(12, 12, w_disc=0.01)
    assert not rule(8, 8, w_lpips=10.0) and rule(8, 8, lanes=2, w_lpips=10.0) and not rule(8, 8, lanes=1)
    # the MinibatchStd claim itself: groups of the reference's reshape(G, -1, ...) against the groups inside the even / odd halves
    for n in (8, 16, 24):
        g = min(4, n)
        groups = {frozenset(m + j * (n // g) for j in range(g)) for m in range(n // g)}
        for half in (range(0, n, 2), range(1, n, 2)):
            half = list(half)
            gh = min(4, len(half))
            for m in range(len(half) // gh):
                assert frozenset(half[m + j * (len(half) // gh)] for j in range(gh)) in groups


def test_bench_witness_accepts_the_reference_and_rejects_a_wrong_batch():
    """bench.py's correctness witness (the last timed batch against the reference's fixture of the workload) on the host: the reference's
    own float32 result passes its own bound, a result two Adam steps off in a handful of entries or with a scrambled image does not, a
    workload without a fixture yields no verdict."""
    import types
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'fullsize_B.npz'))
    args = types.SimpleNamespace(w_disc=0.0, preset='B', res=256, channel_base=32768, batch=8, latent_steps=20)

    def run(w, img_sub):
        w_aug = torch.tensor(w, dtype=torch.float32)[:, None, :].repeat(1, 14, 1)
        full = torch.zeros([8, 2, 256, 256])
        full[:, :, 2::4, 2::4] = torch.tensor(img_sub, dtype=torch.float32)
        return bench.witness(args, w_aug, {'A': full[:, :1], 'B': full[:, 1:]}, 0, 8)
    ok = run(fx['ref32_w'], fx['ref32_img_sub'])
    assert ok['ok'] is True and abs(ok['w_rms_vs_f64'] - ok['reference_f32_rms_vs_f64']) < 1e-9
    bad_w = fx['ref32_w'].copy()
    bad_w[:, :8] += 0.02
    assert run(bad_w, fx['ref32_img_sub'])['ok'] is False
    assert run(fx['ref32_w'], fx['ref32_img_sub'][::-1].copy())['ok'] is False
    args.batch = 4
    assert bench.witness(args, torch.zeros([4, 14, 512]), {}, 0, 4) is None


def test_kernel_class_map_is_the_profilers():
    """One class map (latentaugment_amd/kernel_classes.py) for bench.py's brackets and the rocprofv3 post-processing: its order is the
    C enum's (csrc/la_common.h), every class is described, and the kernels of the contraction classes land where the library brackets them."""
    import re
    from latentaugment_amd import kernel_classes as kc
    src = open(os.path.join(ROOT, 'latentaugment_amd', 'csrc', 'la_common.h')).read()
    enum = re.search(r'enum \{ (LA_PC_CONV_HALO.*?)LA_PC_NCLASS \}', src, re.S).group(1)
    names = [n.strip() for n in enum.replace('= 0', '').split(',') if n.strip()]
    want = {'LA_PC_CONV_HALO': 'conv_halo', 'LA_PC_CONV_FLAT': 'conv_flat', 'LA_PC_CONV_SPLITK': 'conv_splitk', 'LA_PC_CONV_F32': 'conv_f32',
            'LA_PC_PRESPLIT': 'operand_prep', 'LA_PC_FIR': 'fir', 'LA_PC_SEAM': 'seam_bwd', 'LA_PC_TORGB': 'torgb_fwd', 'LA_PC_BANK': 'bank'}
    assert [want[n] for n in names] == kc.CLASSES and set(kc.CLASS_KERNELS) == set(kc.CLASSES)
    assert kc.class_of('void la_conv_bf16_halo_kernel<128, 16, 3, 5>(LaConvArgs)') == 'conv_halo'
    assert kc.class_of('void la_conv_bf16_kernel<128, true, 16, 3, 1>(LaConvArgs)') == 'conv_splitk'
    assert kc.class_of('void la_conv_bf16_kernel<128, false, 16, 3, 2>(LaConvArgs)') == 'conv_flat'
    assert kc.class_of('la_conv_splitk_finish_kernel<4>') == 'conv_splitk' and kc.class_of('la_xscale_bound_kernel') is None
    assert kc.class_of('la_imgrad_pyramid_kernel(PyrArgs)') == 'fir' and kc.class_of('la_step_tail_kernel') is None
