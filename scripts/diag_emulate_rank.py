"""What ONE rank of an N-rank run pays per batch besides its shard's GPU work: the plugin protocol at the global batch 8 N with the
process group calls patched to a single process (rank 0 of N; the all_gather replicates the local shard).  Prints ms per batch against
the plain single-process batch of 8 -- the difference is host-side cost that grows with N (no real communication in it)."""
import os
import random
import sys
import time
import types

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentaugment_amd import synthetic                                     # noqa: E402
from latentaugment_amd.augments import create_augment                       # noqa: E402
from latentaugment_amd.latent_aug import InMemoryLatentCodes                # noqa: E402
import bench                                                                # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')
dist.init_process_group('gloo')
if N > 1:
    dist.get_world_size = lambda group=None: N
    import latentaugment_amd.latent_aug as la_mod

    def device_gather(local, per, batch, group=None):
        """gather_shards as it runs over RCCL (device buffers, no host staging), the collective replaced by a replication"""
        pad = torch.zeros([per] + list(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
        out = torch.empty([N * per] + list(local.shape[1:]), dtype=local.dtype, device=local.device)
        out.view(N, *pad.shape).copy_(pad.unsqueeze(0).expand(N, *pad.shape))
        return out[:batch]
    la_mod.gather_shards = device_gather

sys.argv = ['bench.py']
args = bench.apply_preset(bench.parse())
gb = args.batch * N
sd, meta = synthetic.make_generator_state_dict(img_resolution=args.res, img_channels=2, channel_base=args.channel_base, seed=0)
W, X = synthetic.make_banks(meta['num_ws'], res=args.res, M_w=args.M_w, M_x=args.M_x)
data = synthetic.make_batch(gb, res=args.res, seed=2)
w0 = synthetic.make_latents(gb, seed=1)
codes = InMemoryLatentCodes({p: w0[i, 0].numpy() for i, p in enumerate(data['A_paths'])})
opt = bench.make_opt(args, 0, gb)
opt.inject = dict(generator=sd, banks={'W': W, 'X': X}, latent_codes=codes, group=None)
aug = create_augment(opt)
random.seed(6)


def one():
    aug.set_input(data); aug.forward(); return aug.get_output()


for _ in range(3):
    one()
torch.cuda.synchronize()
t0 = time.time()
K = 8
for _ in range(K):
    out = one()
torch.cuda.synchronize()
print(f'emulated rank 0 of {N}: global batch {gb}, {1e3 * (time.time() - t0) / K:.2f} ms per batch, output {tuple(out["A"].shape)}', flush=True)
if os.environ.get('LA_HOST_PROFILE'):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(4):
        one()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
