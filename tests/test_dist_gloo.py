"""The N>1 path on CPU: world_size-2 (and 3) gloo runs of the tile shard + single gather +
un-interleave used by bench.py --gpus N (yocto-hair_amd/python/yhair_dist.py). No GPU needed:
the per-rank payload is produced by the host restatement of k_pack from a known image."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python"))
import yhair_dist  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    image = rng.uniform(0, 1, (h, w, 4)).astype(np.float32)  # same on every rank
    packed = torch.from_numpy(yhair_dist.pack_tiles_host(image, rank, world))
    assert packed.shape[0] == yhair_dist.shard_pixels(w, h, rank, world)
    dist.barrier()
    full = yhair_dist.gather_framebuffer(packed, w, h, rank, world)
    if rank == 0:
        np.save(out_path, np.stack([full.numpy(), image]))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 72, 40), (3, 37, 21), (2, 8, 8)])
def test_gather_framebuffer_gloo(tmp_path, world, w, h):
    import torch.multiprocessing as mp
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    got, want = np.load(out)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("w,h", [(720, 720), (1280, 1280)])
def test_gather_framebuffer_gloo_world_8_at_the_configs_image_sizes(tmp_path, w, h):
    """The 8-rank gather of BASELINE.json's 8-GPU configs on CPU (gloo): 720^2 has 8 100 tiles — shards of 1 013 and
    1 012 tiles, so the padded capacity and the per-rank trim are exercised — 1280^2 has 25 600 (equal shards)."""
    import torch.multiprocessing as mp
    sizes = [yhair_dist.shard_pixels(w, h, r, 8) for r in range(8)]
    assert (len(set(sizes)) > 1) == (w == 720)
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(8, _free_port(), w, h, out), nprocs=8, join=True)
    got, want = np.load(out)
    assert np.array_equal(got, want)


def test_shards_partition_the_image():
    for w, h in ((720, 720), (720, 405), (13, 7)):
        tx, ty = yhair_dist.tiles_xy(w, h)
        for world in (1, 2, 3, 8):
            tiles = np.concatenate([yhair_dist.shard_tiles(w, h, r, world) for r in range(world)])
            assert sorted(tiles.tolist()) == list(range(tx * ty))
            sizes = [yhair_dist.shard_pixels(w, h, r, world) for r in range(world)]
            assert max(sizes) - min(sizes) <= 64  # interleaving balances the shards


def test_pack_unpack_roundtrip_host():
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 1, (21, 37, 4)).astype(np.float32)
    out = np.zeros_like(img)
    for r in range(3):
        yhair_dist.unpack_tiles_host(yhair_dist.pack_tiles_host(img, r, 3), r, 3, out)
    assert np.array_equal(out, img)


def test_bare_bench_gpus_2_starts_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` without a launcher: the script starts the two ranks itself. Here (no GPU) both
    ranks fail at yh_create — there is no CPU fallback — and the parent must exit non-zero without printing a line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--scale", "0.02", "--resolution", "32", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks run (tests/test_gpu_parity.py covers that)")
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "no HIP device" in out.stderr or "HIP" in out.stderr
