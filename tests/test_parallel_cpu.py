"""CPU, world_size 2, gloo: the frame-sharding exchange schedule of univst_amd.parallel (the same TorchDistComm
object the RCCL path uses) reproduces the UNSHARDED oracle: 5-D GroupNorm through all-reduced partial sums,
sparse-causal attention through a 1-hop halo + rank-0 broadcast, latent_adain through all-reduced content sums."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle import unet_ref


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from univst_amd.parallel import FrameShard, TorchDistComm
        comm = TorchDistComm()
        B, Fr, C, H, G, heads = 3, 4, 32, 4, 8, 2
        N = H * H
        sh = FrameShard(rank, world, Fr, comm=comm)
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, C, Fr, H, H, generator=g) * 2 + 0.3
        gam, bet = torch.randn(C, generator=g), torch.randn(C, generator=g)
        # ---- coupling 1: 5-D GroupNorm statistics
        xl = sh.slice_frames(x)
        cpg = C // G
        part = torch.stack([xl.view(B, G, -1).sum(-1), (xl.view(B, G, -1) ** 2).sum(-1)], -1).reshape(-1).contiguous()
        comm.all_reduce_sum(part)
        part = part.view(B, G, 2)
        cnt = Fr * H * H * cpg
        mean = part[..., 0] / cnt
        var = part[..., 1] / cnt - mean ** 2
        yl = (xl.view(B, G, -1) - mean[..., None]) / torch.sqrt(var[..., None] + 1e-5)
        yl = yl.view_as(xl) * gam[None, :, None, None, None] + bet[None, :, None, None, None]
        ref = sh.slice_frames(F.group_norm(x, G, gam, bet, 1e-5))
        assert (yl - ref).abs().max() < 1e-4
        # ---- coupling 2 (round 6 schedule): sparse-causal attention; what travels is the boundary frames' HIDDEN rows ([B, N, C]: half a K|V pack), the
        # receiver runs norm1 -> to_k | to_v (and, inside the PnP window, the per-frame AdaIN shift: pnp_utils.py:44-57,114-125) on the two halo frames
        # itself, and the attention runs in two phases: the key frames the rank holds first, the halo frames continue from the (m, l, o) softmax state
        hid = torch.randn(B * Fr, N, C, generator=g)
        lg, lb = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.1
        wq, wk, wv = (torch.randn(C, C, generator=g) / C ** 0.5 for _ in range(3))
        ln = lambda t: F.layer_norm(t, (C,), lg, lb, 1e-5)
        proj = lambda t: (ln(t) @ wq.T, ln(t) @ wk.T, ln(t) @ wv.T)
        d = C // heads

        def phase(qf, ks, vs):          # softmax state of one frame's queries over a key list: (m, l, o normalised) per head; empty list -> l = 0
            if not ks:
                return None
            kk, vv = torch.cat(ks, 1), torch.cat(vs, 1)
            sc = torch.einsum("bqhd,bkhd->bhqk", qf.view(B, N, heads, d), kk.view(B, -1, heads, d)) / d ** 0.5
            m = sc.amax(-1)
            pr = torch.exp(sc - m[..., None])
            l = pr.sum(-1)
            o = torch.einsum("bhqk,bkhd->bqhd", pr / l[..., None], vv.view(B, -1, heads, d))
            return m, l, o

        def merge(s1, s2):               # csrc/attention.hip attn_merge_coef
            if s1 is None:
                return s2[2]
            m = torch.maximum(s1[0], s2[0])
            a1, a2 = s1[1] * torch.exp(s1[0] - m), s2[1] * torch.exp(s2[0] - m)
            w1, w2 = (a1 / (a1 + a2)).transpose(1, 2)[..., None], (a2 / (a1 + a2)).transpose(1, 2)[..., None]
            return s1[2] * w1 + s2[2] * w2

        for index, idx in (([-1, 0, "first"], None), ([-1, "first"], 12), ([-1, "first"], 40)):
            q, k, v = proj(hid)
            if idx is not None:           # PnP layers: the shift acts on all frames of the three branches before the gather
                q, k, v = unet_ref.pnp_shift(q, k, v, idx)
            full = unet_ref.sdpa(q, unet_ref.sparse_causal_gather(k, Fr, index), unet_ref.sparse_causal_gather(v, Fr, index), heads)
            loc = lambda t: t.view(B, Fr, N, C)[:, sh.f0:sh.f0 + sh.local]
            hl = loc(hid)
            send_last, first = hl[:, sh.local - 1].contiguous(), hl[:, 0].contiguous()
            recv_prev, recv_first = torch.zeros_like(send_last), torch.zeros_like(first)
            comm.halo_and_broadcast(send_last, first, recv_prev, recv_first)
            ql, kl, vl = proj(hl.reshape(B * sh.local, N, C))
            if idx is not None:
                ql, kl, vl = unet_ref.pnp_shift(ql, kl, vl, idx)
            ql, kl, vl = (t.view(B, sh.local, N, C) for t in (ql, kl, vl))
            halo = {}
            if rank > 0:                  # the two halo frames of every branch: projected (and shifted) as two one-frame clips
                for name, pack in (("prev", recv_prev), ("first", recv_first)):
                    qh, kh, vh = proj(pack)
                    if idx is not None:
                        qh, kh, vh = unet_ref.pnp_shift(qh, kh, vh, idx)
                    halo[name] = (kh, vh)
            outs = []
            for f in range(sh.local):
                lk, lv, hk, hv = [], [], [], []
                if f > 0 or rank == 0:    # previous frame held locally (rank 0: frame 0's predecessor clips to itself)
                    lk.append(kl[:, max(f - 1, 0)]); lv.append(vl[:, max(f - 1, 0)])
                else:
                    hk.append(halo["prev"][0]); hv.append(halo["prev"][1])
                if 0 in index:
                    lk.append(kl[:, f]); lv.append(vl[:, f])
                if rank == 0:
                    lk.append(kl[:, 0]); lv.append(vl[:, 0])
                else:
                    hk.append(halo["first"][0]); hv.append(halo["first"][1])
                s1 = phase(ql[:, f], lk, lv)
                o = merge(s1, phase(ql[:, f], hk, hv)) if hk else s1[2]
                outs.append(o.reshape(B, N, C))
            got = torch.stack(outs, 1)
            assert (got - loc(full)).abs().max() < 2e-5, (index, idx, float((got - loc(full)).abs().max()))
        # ---- coupling 3: latent_adain content statistics; gather
        c5, s5 = torch.randn(1, 4, Fr, H, H, generator=g), torch.randn(1, 4, Fr, H, H, generator=g) * 0.5 + 0.2
        cl, sl = sh.slice_frames(c5), sh.slice_frames(s5)
        st = torch.stack([cl.sum(dim=(0, 2, 3, 4)), (cl ** 2).sum(dim=(0, 2, 3, 4))], -1).reshape(-1).contiguous()
        comm.all_reduce_sum(st)
        st = st.view(4, 2)
        n = Fr * H * H
        mu = st[:, 0] / n
        rstd = 1 / torch.sqrt(st[:, 1] / n - mu ** 2 + 1e-5)
        smu, sstd = sl.mean(dim=(0, 3, 4), keepdim=True), sl.std(dim=(0, 3, 4), keepdim=True)
        got = (cl - mu[None, :, None, None, None]) * rstd[None, :, None, None, None] * sstd + smu
        assert (got - sh.slice_frames(unet_ref.latent_adain(c5, s5))).abs().max() < 1e-4
        assert torch.equal(sh.gather_frames(sh.slice_frames(c5)), c5)
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_frame_shard_schedule_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def _sd3_worker(rank, world, port, out):
    """the SD3 / SD3.5 exchange schedule of csrc/sd3.hip (round 6) restated with torch ops and the gloo communicator, against the UNSHARDED oracle
    (oracle/sd3_ref.joint_attention = backbones/video_diffusion_sd3/pnp_utils.py:17-271): the previous frame travels as the layer's hidden rows (the
    receiver projects, RMS-normalises and shifts it), the clip's first frame as the K | V rows rank 0 finished, and the joint attention runs in two
    phases — keys the rank holds ++ the text keys, then the halo frames — for the image AND the text queries."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import sd3_ref
        from univst_amd.parallel import Sd3FrameShard, TorchDistComm
        comm = TorchDistComm()
        B, Fr, N, Nt, C, heads = 3, 4, 6, 3, 16, 2
        d = C // heads
        sh = Sd3FrameShard(rank, world, Fr, comm=comm)
        g = torch.Generator().manual_seed(5)
        rn = lambda *s_: torch.randn(*s_, generator=g)
        P = {}
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            P[n + ".weight"], P[n + ".bias"] = rn(C, C) / C ** 0.5, rn(C) * 0.1
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            P[n + ".weight"] = 1.0 + 0.2 * rn(d)
        hidden, enc = rn(B * Fr, N, C), rn(B * Fr, Nt, C)
        lin = lambda x, n: F.linear(x, P[n + ".weight"], P[n + ".bias"])
        sp = lambda t: t.view(t.shape[0], -1, heads, d).transpose(1, 2)                     # [(b f), heads, tokens, d]
        rms = lambda x, n: sd3_ref._rms(x, P[n + ".weight"], 1e-6)

        def shifted(q, k, v, idx):              # pnp_utils.py:183-194 on [(3 c), heads, N, d]: per frame, so it applies to any subset of whole frames
            c = q.shape[0] // 3
            if not (idx >= 0 and idx <= 0.6 * 50):
                return q, k, v
            beta = (0.9 - 0.1) / (0 - 0.6 * 50) * (idx - 0.6 * 50) + 0.1
            q, k, v = q.clone(), k.clone(), v.clone()
            q[2 * c:] = 2.0 * (0.8 * q[:c] + 0.2 * q[2 * c:])
            k[2 * c:] = beta * sd3_ref.attention_adain(k[2 * c:], k[c:2 * c]) + (1 - beta) * k[c:2 * c]
            v[2 * c:] = beta * sd3_ref.attention_adain(v[2 * c:], v[c:2 * c]) + (1 - beta) * v[c:2 * c]
            return q, k, v

        def phase(qf, ks, vs):                  # qf [3, heads, nq, d]; key lists of [3, heads, n, d]
            if not ks:
                return None
            kk, vv = torch.cat(ks, 2), torch.cat(vs, 2)
            sc = qf @ kk.transpose(2, 3) / d ** 0.5
            m = sc.amax(-1)
            pr = torch.exp(sc - m[..., None])
            l = pr.sum(-1)
            return m, l, (pr / l[..., None]) @ vv

        def merge(s1, s2):
            if s2 is None:
                return s1[2]
            m = torch.maximum(s1[0], s2[0])
            a1, a2 = s1[1] * torch.exp(s1[0] - m), s2[1] * torch.exp(s2[0] - m)
            return s1[2] * (a1 / (a1 + a2))[..., None] + s2[2] * (a2 / (a1 + a2))[..., None]

        for idx in (12, 45):
            want_i, want_t = sd3_ref.joint_attention(P, heads, hidden, enc, idx=idx, shift=True, clip_length=Fr)
            loc5 = lambda t: t.view(B, Fr, *t.shape[1:])[:, sh.f0:sh.f0 + sh.local]
            hl, el = loc5(hidden), loc5(enc)                                               # [3, local, tokens, C]
            flat = lambda t: t.reshape(B * sh.local, *t.shape[2:])
            q, k, v = sp(lin(flat(hl), "to_q")), sp(lin(flat(hl), "to_k")), sp(lin(flat(hl), "to_v"))
            q, k = rms(q, "norm_q"), rms(k, "norm_k")
            q, k, v = shifted(q, k, v, idx)
            eq, ek, ev = sp(lin(flat(el), "add_q_proj")), sp(lin(flat(el), "add_k_proj")), sp(lin(flat(el), "add_v_proj"))
            eq, ek = rms(eq, "norm_added_q"), rms(ek, "norm_added_k")
            v5 = lambda t: t.view(B, sh.local, *t.shape[1:])                               # [3, local, heads, tokens, d]
            q5, k5, v5_, eq5, ek5, ev5 = (v5(t) for t in (q, k, v, eq, ek, ev))
            # ---- the two packs: previous frame = hidden rows [3, N, C]; first frame = finished K | V [3, N, 2C] (rank 0's)
            send_last = hl[:, sh.local - 1].contiguous()
            kv_first = torch.cat([k5[:, 0].transpose(1, 2).reshape(B, N, C), v5_[:, 0].transpose(1, 2).reshape(B, N, C)], -1).contiguous()
            recv_prev, recv_first = torch.zeros_like(send_last), torch.zeros_like(kv_first)
            dist.broadcast(kv_first if rank == 0 else recv_first, src=0)
            if rank < world - 1:
                dist.send(send_last, rank + 1)
            if rank > 0:
                dist.recv(recv_prev, rank - 1)
                pq, pk, pv = sp(lin(recv_prev, "to_q")), sp(lin(recv_prev, "to_k")), sp(lin(recv_prev, "to_v"))
                pk = rms(pk, "norm_k")
                _, pk, pv = shifted(pq, pk, pv, idx)                                       # one frame of the three branches: c = 1
                fk, fv = sp(recv_first[..., :C]), sp(recv_first[..., C:])
            oi, ot = [], []
            for f in range(sh.local):
                if rank == 0:                   # everything local; duplicates as the reference lists them ['first', f - 1 clipped, f]
                    lk, lv = [k5[:, 0], k5[:, max(f - 1, 0)], k5[:, f]], [v5_[:, 0], v5_[:, max(f - 1, 0)], v5_[:, f]]
                    hk, hv = [], []
                else:
                    lk, lv = ([k5[:, f - 1]] if f > 0 else []) + [k5[:, f]], ([v5_[:, f - 1]] if f > 0 else []) + [v5_[:, f]]
                    hk, hv = [fk] + ([pk] if f == 0 else []), [fv] + ([pv] if f == 0 else [])
                lk, lv = lk + [ek5[:, f]], lv + [ev5[:, f]]                                # the text keys ride in the first phase
                for qq, dst in ((q5[:, f], oi), (eq5[:, f], ot)):
                    o = merge(phase(qq, lk, lv), phase(qq, hk, hv))
                    dst.append(o.transpose(1, 2).reshape(B, -1, C))
            got_i = lin(torch.stack(oi, 1), "to_out.0")
            got_t = lin(torch.stack(ot, 1), "to_add_out")
            ei, et = (got_i - loc5(want_i)).abs().max(), (got_t - loc5(want_t)).abs().max()
            assert ei < 2e-5 and et < 2e-5, (idx, float(ei), float(et))
        out.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sd3_frame_shard_schedule_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sd3_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_frame_shard_slicing():
    from univst_amd.parallel import FrameShard
    t = torch.arange(2 * 4 * 8).view(1, 2, 8, 2, 2)
    parts = [FrameShard(r, 4, 8).slice_frames(t) for r in range(4)]
    assert torch.equal(torch.cat(parts, 2), t)
    with pytest.raises(ValueError):
        FrameShard(0, 3, 16)


def test_sd3_frame_shard_slicing():
    """SD3 path: frames are the batch axis and the three branches are concatenated along it; a rank holds the same frame range of
    every branch.  (The K/V exchange itself lives in the native library and is tested by two processes on a GPU.)"""
    from univst_amd.parallel import Sd3FrameShard
    F_, W = 16, 4
    t = torch.arange(3 * F_ * 2).view(3 * F_, 2)
    parts = [Sd3FrameShard(r, W, F_).slice_branches(t) for r in range(W)]
    assert all(p.shape == (3 * F_ // W, 2) for p in parts)
    # re-assemble branch by branch: [branch][rank][local frame] -> the original order
    back = torch.cat([torch.cat([p.chunk(3)[b] for p in parts]) for b in range(3)])
    assert torch.equal(back, t)
    sh = Sd3FrameShard(2, W, F_)
    assert (sh.f0, sh.local) == (8, 4) and torch.equal(sh.slice_frames(t[:F_]), t[8:12])
    assert Sd3FrameShard(0, 1, F_).gather_frames(t) is t
    with pytest.raises(ValueError):
        Sd3FrameShard(0, 3, 16)


def test_init_distributed_without_a_launcher_is_a_no_op(monkeypatch):
    """a plain `python run_video_style_transfer_sd.py` (no RANK / WORLD_SIZE): nothing is initialised, the pipeline stays on one GPU"""
    import torch.distributed as dist
    from univst_amd.parallel import init_distributed, dist_rank_world
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert init_distributed() == (0, 1)
    assert not dist.is_initialized() and dist_rank_world() == (0, 1)


def _init_rank(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        import torch.distributed as dist
        from univst_amd.parallel import init_distributed, dist_rank_world, FrameShard
        got = init_distributed(backend="gloo", timeout_s=60)
        again = init_distributed(backend="gloo")              # idempotent (the CLI and a caller may both do it)
        sh = FrameShard(*dist_rank_world(), frames=8)
        q.put((rank, got, again, dist_rank_world(), (sh.f0, sh.local), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:       # noqa: BLE001
        import traceback
        q.put((rank, None, None, None, None, traceback.format_exc()))


def test_init_distributed_from_the_launcher_environment_world2():
    """what `torchrun --nproc-per-node 2 src/sd/run_video_style_transfer_sd.py` gives the script: RANK / WORLD_SIZE / LOCAL_RANK ->
    a process group (gloo here: no GPU) and this rank's frame range"""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_init_rank, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
    for rank, got, again, rw, rng, err in res:
        assert err is None, err
        assert got == (rank, 2) and again == (rank, 2) and rw == (rank, 2) and rng == (rank * 4, 4)


def test_ipc_bringup_agrees_on_failure_before_any_device_collective():
    """a rank whose univst_comm_create / hipIpcOpenMemHandle fails must not leave its peers spinning in the device-side
    self-test.  On this GPU-less box univst_comm_create fails on every rank: each learns of the others' failure through the host-side
    exchange (here two threads standing in for two processes) and ALL raise the same RuntimeError naming the failed ranks — the point
    at which FrameShard.attach falls back to the torch.distributed callbacks on every rank together."""
    import threading
    from univst_amd import parallel
    world = 2
    bar = threading.Barrier(world)
    box = [None] * world
    out = {}

    def exchange_for(rank):
        def exchange(blob):
            box[rank] = blob
            bar.wait(timeout=60)
            got = list(box)
            bar.wait(timeout=60)
            return got
        return exchange

    def run(rank):
        try:
            parallel.NativeIpcComm(rank, world, 1 << 20, device=torch.device("cpu"), exchange=exchange_for(rank))
            out[rank] = "constructed"
        except RuntimeError as e:
            out[rank] = str(e)
        except Exception as e:       # noqa: BLE001
            out[rank] = f"{type(e).__name__}: {e}"

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert set(out) == {0, 1}, out
    for r in range(world):
        assert "IPC communicator: create/export failed on rank(s) [0, 1]" in out[r], out[r]
