"""Frame sharding of the three-branch loop over the GPUs of one node: one process per GPU, RCCL over xGMI through
``torch.distributed`` (backend "nccl" == RCCL on ROCm).

Rank r owns frames [r*Fl, (r+1)*Fl) of ALL THREE branches, so the PnP injection (content/style -> stylised
branch) and every per-frame op stay local.  Three couplings cross ranks:

  1. 5-D GroupNorm (45 per UNet call): statistics span all frames of a branch -> the native graph reduces its
     partial sums to [B, groups, 2] fp32 (768 B) and calls back ``allreduce`` (SUM, in place).
  2. sparse-causal attn1 (16 per UNet call): frame f attends to {f-1, (f), 0}.  The previous frame of a rank's
     first local frame lives on rank-1 and frame 0 on rank 0 -> a 1-hop halo send/recv plus a broadcast from
     rank 0 of one [B, N, 2C] K|V pack each (not an all-gather of all K/V: 8x fewer bytes on the per-link-bound
     xGMI ring).
  3. latent_adain (1 + 5 calls per run): content statistics over (F,h,w) -> all-reduce of [C, 2] fp32.

Two interchangeable communicators sit behind the UNet graph's two hooks:

  * ``NativeIpcComm`` (default on one node): the library's own communicator (csrc/comm.hip, ``univst_comm_*``) — peers' regions
    mapped through HIP IPC, every coupling a device-side peer write + flag on the UNet's stream.  No host callback runs during a
    forward; ``torch.distributed`` is used once, to all-gather the IPC handles (and for the final frame all-gather).
  * ``TorchDistComm`` / ``HostStagedDistComm``: host callbacks that enqueue RCCL / gloo collectives through ``torch.distributed``
    (the round-1/2 path; kept as the fallback when IPC mapping fails, e.g. across nodes, and for CPU-side tests).

Tests swap in ``ThreadLoopbackComm`` (ranks = host threads sharing one GPU) to validate the exchange schedule on a 1-GPU box, a
gloo-backed CPU instance to validate the sharded algorithm against the unsharded oracle, and two PROCESSES sharing the GPU for both
communicators.
"""
import os
import ctypes as C
import threading
from typing import List, Optional

import torch

from . import _native


def init_distributed(backend: Optional[str] = None, timeout_s: int = 300):
    """One process per GPU, started by ``torchrun --nproc-per-node N`` (or ``python -m torch.distributed.run``): read
    RANK / WORLD_SIZE / LOCAL_RANK, bind this process to its GPU and join the process group (backend "nccl" = RCCL on ROCm;
    ``UNIVST_DIST_BACKEND`` / ``backend`` override it, e.g. "gloo" when several ranks share one GPU during bring-up).
    Returns (rank, world).  Without the launcher environment (a plain ``python run_video_style_transfer_sd.py``): (0, 1) and
    nothing is initialised — the single-GPU path is untouched."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import datetime
    import torch.distributed as dist
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(min(local, torch.cuda.device_count() - 1))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or os.environ.get("UNIVST_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"timeout": datetime.timedelta(seconds=timeout_s)}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend, **kw)
    return rank, world


def dist_rank_world():
    """(rank, world) of the default process group; (0, 1) when torch.distributed is not initialised."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:       # noqa: BLE001
        pass
    return 0, 1


def auto_frame_shard(pipe, frames: int, latent_hw, check_inputs=None, branches: int = 3, log=print):
    """The frame shard of ``pipe.unet`` for the current process group, built (and checked) once per pipeline and geometry: None
    when there is one process; otherwise a ``FrameShard`` attached to the UNet whose communicator passed ``self_check`` (library IPC
    first, ``torch.distributed`` callbacks as the fall-back).  ``check_inputs`` = (content [1,4,F,h,w], style, text3) for the check."""
    rank, world = dist_rank_world()
    if world <= 1:
        return None
    key = (rank, world, frames, tuple(latent_hw))
    cache = pipe.__dict__.setdefault("_univst_shards", {})
    if key in cache:
        return cache[key]
    h, w = latent_hw
    shard = FrameShard(rank, world, frames)
    shard.attach(pipe.unet, max_tokens=h * w, branches=branches)
    if check_inputs is not None:
        rep = shard.self_check(pipe, *check_inputs)
        if rank == 0 and log is not None:
            log(f"[univst_amd] frame shard over {world} GPUs: {rep}")
    cache[key] = shard
    pipe.shard_report = getattr(shard, "report", None)
    return shard


class FrameShard:
    def __init__(self, rank: int, world: int, frames: int, comm=None):
        if frames % world != 0:
            raise ValueError(f"frames={frames} must be divisible by the number of GPUs ({world})")
        self.rank, self.world, self.frames = rank, world, frames
        self.local = frames // world
        self.f0 = rank * self.local
        self.comm = comm
        self._keep = []          # ctypes callbacks + workspace must outlive the native handle
        self.ws = None

    # ------------------------------------------------------------------ data layout
    def slice_frames(self, t: torch.Tensor) -> torch.Tensor:
        """[.., .., F, h, w] -> this rank's frames (contiguous)."""
        if self.world == 1:
            return t
        return t[:, :, self.f0:self.f0 + self.local].contiguous()

    def gather_frames(self, t: torch.Tensor) -> torch.Tensor:
        """inverse of slice_frames on every rank (final latents -> full clip)."""
        if self.world == 1:
            return t
        return torch.cat(self.comm.all_gather(t.contiguous()), dim=2)

    # ------------------------------------------------------------------ native hooks
    def attach(self, unet, max_tokens: int = 4096, max_channels: int = None, branches: int = 3):
        """register the comm callbacks + workspace on the UNet's native handle (no-op for world == 1).  ``max_tokens`` is
        only the initial size: UNet.forward calls ``ensure`` with the real latent size and the workspace grows on demand;
        a rebuilt native handle (``.half()``, ``load_state_dict`` ...) is re-registered the same way."""
        if self.world == 1:
            return
        self._branches, self._max_channels = branches, max_channels
        if self.comm is None:
            import torch.distributed as dist
            kind = os.environ.get("UNIVST_COMM", "ipc")
            if kind == "ipc":
                try:
                    self.comm = NativeIpcComm(self.rank, self.world, self._ws_bytes(unet, max_tokens), device=unet.device)
                except Exception as e:       # every rank fails or none does (the self-test is a collective): fall back together
                    if self.rank == 0:
                        print(f"[univst_amd] native IPC communicator unavailable ({type(e).__name__}: {e}); using torch.distributed callbacks", flush=True)
                    self.comm = None
            if self.comm is None:
                self.comm = TorchDistComm() if dist.get_backend() == "nccl" else HostStagedDistComm()
        self._tokens, self._handle = 0, None
        self.error = None
        unet._frame_shard = self
        unet._sync_native()
        self.ensure(unet, max_tokens)

    def ensure(self, unet, tokens: int):
        """called by UNet.forward before every native call: the comm hooks are on the CURRENT native handle and the
        workspace holds the largest K|V pack of a ``tokens``-token latent."""
        if self.world == 1:
            return
        handle = unet._native_handle
        if handle is self._handle and tokens <= self._tokens:
            return
        tokens = max(tokens, self._tokens)
        if isinstance(self.comm, (NativeIpcComm, EmulatedIpcComm)):     # the communicator owns the (IPC-shared) workspace; no callbacks
            self.comm.ensure_bytes(self._ws_bytes(unet, tokens))
            _native.check(_native.load().univst_unet_set_comm_native(handle, self.comm.ptr), "unet_set_comm_native")
            self._tokens, self._handle = tokens, handle
            return
        nbytes = self._ws_bytes(unet, tokens, slots=4)
        if self.ws is None or self.ws.numel() < nbytes:
            self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=unet.device)
        nbytes = self.ws.numel()
        shard = self

        def allreduce(user, byte_off, count):
            try:
                shard.comm.all_reduce_sum(shard.ws[byte_off:byte_off + 4 * count].view(torch.float32))
                return 0
            except BaseException as e:      # never unwind through the C frames: report, the native call returns an error
                shard.error = e
                return 1

        def kv_exchange(user, off_send, off_first, off_prev, off_rfirst, nb):
            try:
                w = shard.ws
                shard.comm.halo_and_broadcast(w[off_send:off_send + nb], w[off_first:off_first + nb], w[off_prev:off_prev + nb],
                                              w[off_rfirst:off_rfirst + nb])
                return 0
            except BaseException as e:
                shard.error = e
                return 1

        ar, kv = _native.ALLREDUCE_FN(allreduce), _native.KVEXCHANGE_FN(kv_exchange)
        self._keep = [ar, kv]
        _native.check(_native.load().univst_unet_set_comm(handle, self.rank, self.world, self.ws.data_ptr(), nbytes, ar, kv, None),
                      "unet_set_comm")
        self._tokens, self._handle = tokens, handle

    # ------------------------------------------------------------------ the sharded path checked against the unsharded one
    def detached(self, unet):
        """context manager: ``unet`` runs UNSHARDED (world 1, whole clips) inside the block, e.g. as the reference of
        ``self_check``; the hooks are put back on exit."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            if self.world == 1:
                yield
                return
            unet._sync_native()
            unet._frame_shard = None
            _native.check(_native.load().univst_unet_set_comm(unet._native_handle, 0, 1, None, 0, _native.ALLREDUCE_FN(), _native.KVEXCHANGE_FN(), None), "unet_set_comm(detach)")
            try:
                yield
            finally:
                unet._frame_shard = self
                self._handle = None          # re-register on the next forward
                self.ensure(unet, max(self._tokens, 1))
        return cm()

    def fresh_reference(self, unet):
        """context manager: inside the block ``unet`` runs UNSHARDED on a SEPARATELY BUILT native handle (its own weight copy, derived tensors, source
        tables and arena; no communicator ever attached) — the reference of ``self_check``: an error that depends on the rank or on state of the sharded
        handle cannot cancel against itself.  The sharded handle is put back, untouched, on exit and the fresh one destroyed."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            if self.world == 1:
                yield
                return
            unet._sync_native()
            keys = ("_native_handle", "_native_sum", "_native_fp", "_native_dirty")
            saved = {k: unet.__dict__.get(k) for k in keys}
            unet._native_handle, unet._native_dirty, unet._frame_shard = None, True, None
            try:
                yield
            finally:
                torch.cuda.synchronize()
                fresh = unet.__dict__.get("_native_handle")
                if fresh is not None and fresh is not saved["_native_handle"]:
                    _native.load().univst_unet_destroy(fresh)
                for k in keys:
                    setattr(unet, k, saved[k])
                unet._frame_shard = self
        return cm()

    def self_check(self, pipe, content_full, style_full, text3, idx: int = 10, t: int = 781, tol: float = 2e-2) -> dict:
        """Every rank compares ONE frame-sharded three-branch forward inside the PnP window (K/V exchange + all 45 GroupNorm
        all-reduces) with the unsharded forward of a SEPARATELY BUILT native handle of the same weights on the whole clip (``fresh_reference``),
        before anything that matters runs through the communicator.  A communicator that fails — wrong numbers, a refused IPC mapping, a bounded wait that gave up — is replaced
        (library IPC -> torch.distributed callbacks) on ALL ranks together; if that one fails too a RuntimeError carries both errors.
        Returns (and keeps in ``self.report``) {"comm", "max_rel_err_vs_unsharded"[, "rejected"]}.  A collective: call on all ranks."""
        if self.world == 1:
            self.report = {"comm": None, "max_rel_err_vs_unsharded": 0.0}
            return self.report
        import torch.distributed as dist
        from .backbones.video_diffusion_sd.pnp_utils import register_time
        unet = pipe.unet
        registered = _pnp_registered(unet)
        mix = (0.5 * (content_full.float() + style_full.float())).to(torch.float16)
        xf = torch.cat([content_full.to(torch.float16), style_full.to(torch.float16), mix]).contiguous()
        text3 = text3.to(torch.float16).contiguous()
        with self.fresh_reference(unet):
            if registered:
                register_time(pipe, idx)
            want = unet(xf, t, encoder_hidden_states=text3).sample[:, :, self.f0:self.f0 + self.local].float()

        def once():
            try:
                if registered:
                    register_time(pipe, idx)
                got = unet(self.slice_frames(xf), t, encoder_hidden_states=text3).sample.float()
                torch.cuda.synchronize()
                err = float((got - want).abs().max() / want.abs().max())
            except Exception as e:      # noqa: BLE001  (a failed collective surfaces as RuntimeError out of the native call)
                err = float("inf")
                print(f"[univst_amd] rank {self.rank}: sharded self-check raised {type(e).__name__}: {e}", flush=True)
            errs = [None] * self.world
            dist.all_gather_object(errs, err)
            return max(errs)

        err = once()
        kind = type(self.comm).__name__
        self.report = {"comm": kind, "max_rel_err_vs_unsharded": err}
        force = os.environ.get("UNIVST_SHARD_REJECT_FIRST") == "1"       # test aid: walk the fall-back (collective close of the rejected communicator) on a healthy box
        if force or not err < tol:
            if self.rank == 0:
                print(f"[univst_amd] frame-sharded forward through {kind} differs from the unsharded one (max rel err {err}); "
                      "switching to the torch.distributed callbacks", flush=True)
            rejected = self.comm
            unet._sync_native()                            # the UNet lets go of the rejected communicator's regions ...
            _native.check(_native.load().univst_unet_set_comm(unet._native_handle, 0, 1, None, 0, _native.ALLREDUCE_FN(), _native.KVEXCHANGE_FN(), None),
                          "unet_set_comm(detach)")
            if hasattr(rejected, "close"):
                rejected.close()                           # ... before they are unmapped, on all ranks together
            self.comm = TorchDistComm() if dist.get_backend() == "nccl" else HostStagedDistComm()
            self._handle, self._tokens, self.ws = None, 0, None
            self.ensure(unet, xf.shape[-1] * xf.shape[-2])
            err2 = once()
            self.report = {"comm": type(self.comm).__name__, "max_rel_err_vs_unsharded": err2, "rejected": {"comm": kind, "max_rel_err": err}}
            if not err2 < tol:
                raise RuntimeError(f"frame-sharded forward is wrong through both communicators ({kind}: {err}, {type(self.comm).__name__}: {err2})")
        return self.report

    def _ws_bytes(self, unet, tokens: int, slots: int = 6) -> int:
        """64 KiB (GroupNorm partials) + `slots` K|V packs of a `tokens`-token latent: the largest pack is B * N * 2C fp16 over the
        attention levels (N shrinks 4x per level while C grows <= 2x)."""
        boc = unet.config.block_out_channels
        pack = max(self._branches * (tokens >> (2 * i)) * 2 * c * 2 for i, c in enumerate(boc[:3]))
        return 65536 + slots * ((pack + 255) // 256 * 256) + 4096

    def raise_pending(self, what: str):
        """a collective failed inside a native call: surface the original exception on THIS rank right away (the process
        then dies and the launcher tears the job down) instead of leaving the peers blocked behind a swallowed error."""
        if getattr(self, "error", None) is not None:
            e, self.error = self.error, None
            abort = getattr(self.comm, "abort", None)
            if abort is not None:
                abort()
            raise RuntimeError(f"{what}: collective failed on rank {self.rank}/{self.world}") from e

    # ------------------------------------------------------------------ sharded latent_adain (pnp_utils.py:128-139)
    def latent_adain(self, cnt: torch.Tensor, sty: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return _native.latent_adain(cnt, sty)
        lib = _native.load()
        _, Cl, Fl, H, W = cnt.shape
        st = torch.empty(Cl * 2, device=cnt.device, dtype=torch.float32)
        _native.check(lib.univst_latent_adain_stats(cnt.data_ptr(), st.data_ptr(), Cl, Fl, H * W, _native.stream_ptr()), "adain_stats")
        self.comm.all_reduce_sum(st)
        out = torch.empty_like(cnt)
        _native.check(lib.univst_latent_adain_apply(cnt.data_ptr(), sty.data_ptr(), st.data_ptr(), self.frames * H * W, out.data_ptr(),
                                                    Cl, Fl, H * W, _native.stream_ptr()), "adain_apply")
        return out

    # ------------------------------------------------------------------ one step of the transfer loop on the local frames
    def make_step_fn(self, pipe, content, style, text3, mask_m=None, n=50):
        from . import engine
        from .backbones.video_diffusion_sd.pnp_utils import register_time
        ts = pipe.scheduler.timesteps

        def step(i, latents):
            i = i % n
            t = ts[i]
            c_t, s_t = content[n - i], style[n - i]
            if mask_m is not None and i <= 0.9 * n:
                latents = _native.mask_blend(latents, c_t, mask_m)
            if i > 0.8 * n and i <= 0.9 * n:
                latents = _native.mask_blend(self.latent_adain(latents, s_t), c_t, mask_m)
            register_time(pipe, i)
            x = torch.cat([c_t, s_t, latents])
            eps = pipe.unet(x, t, encoder_hidden_states=text3).sample[2:3]
            return engine.ddim_step(pipe.scheduler, eps, t, latents)
        return step


def _pnp_registered(unet) -> bool:
    """register_spatial_attention_pnp was applied to this UNet (whether or not register_time has run yet)."""
    from .backbones.video_diffusion_sd.pnp_utils import PNP_LAYERS
    return any(getattr(unet.up_blocks[r].attentions[b].transformer_blocks[0].attn1, "_univst_native_pnp", False)
               for r, bs in PNP_LAYERS.items() for b in bs)


class Sd3FrameShard:
    """Frame shard of the SD3 / SD3.5 path: frames are the batch axis there ([F, C, h, w] per branch), rank r holds frames
    [r*F/W, (r+1)*F/W) of every branch, and the only coupling is the cross-frame key set of the joint attention — K | V of the clip's
    first frame and of the previous frame, exchanged inside ``univst_sd3_joint_attention`` through the library's communicator
    (csrc/comm.hip: peer writes + flags; one exchange and one barrier per attention layer).  AdaIN statistics, the latent AdaIN and
    the Euler step are per frame: local.  IPC communicator only (one node)."""

    def __init__(self, rank: int, world: int, frames: int, comm=None):
        if frames % world:
            raise ValueError(f"{frames} frames do not split over {world} ranks")
        self.rank, self.world, self.frames = rank, world, frames
        self.local = frames // world
        self.f0 = rank * self.local
        self.comm = comm

    def slice_frames(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.f0:self.f0 + self.local].contiguous()

    def slice_branches(self, t: torch.Tensor, branches: int = 3) -> torch.Tensor:
        """[branches*F, ...] -> this rank's frames of every branch, [branches*F/W, ...]"""
        return torch.cat([self.slice_frames(c) for c in t.chunk(branches)])

    def attach(self, model, tokens: int, branches: int = 3):
        """model: the native CustomSD3Transformer2DModel; tokens: image tokens per frame (sizes the K/V packs)."""
        if self.world == 1:
            return self
        inner = model.inner_dim
        pack = (branches * tokens * 2 * inner * 2 + 255) // 256 * 256
        need = 65536 + 6 * pack + 4096
        if self.comm is None:
            self.comm = NativeIpcComm(self.rank, self.world, need, device=model.device)
        else:
            self.comm.ensure_bytes(need)
        for blk in model.transformer_blocks:
            blk.attn._uv_frame_shard = self
            if blk.attn2 is not None:
                blk.attn2._uv_frame_shard = self
        return self

    def gather_frames(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t
        return torch.cat(self.comm.all_gather(t.contiguous()), dim=0)


# ------------------------------------------------------------------------------------------------ communicators
class NativeIpcComm:
    """The library's own communicator (include/univst.h ``univst_comm_*``, csrc/comm.hip): device-side peer writes + flags through
    IPC-mapped fine-grained memory.  ``torch.distributed`` (any backend) only carries the 64-byte handles at bring-up and the final
    frame all-gather; ``exchange`` may replace it (callable: bytes -> list of every rank's bytes)."""

    def __init__(self, rank: int, world: int, ws_bytes: int, device=None, exchange=None):
        self.rank, self.world = rank, world
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._exchange = exchange or self._dist_exchange
        self.ptr, self.ws_bytes = None, 0
        self._build(ws_bytes)

    @staticmethod
    def _dist_exchange(blob: bytes):
        import torch.distributed as dist
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, blob)
        return out

    def _agree(self, ok: bool, what: str, payload: bytes = b""):
        """host-side agreement: every rank learns whether EVERY rank got through `what`; if one did not, all raise
        together — so the fall-back to the torch.distributed callbacks happens on all ranks at once instead of the healthy
        ranks spinning in a device-side collective their failed peer never joins.  Returns the peers' payloads."""
        blobs = self._exchange((b"\x01" if ok else b"\x00") + payload)
        bad = [r for r, b in enumerate(blobs) if not b or b[0] != 1]
        if len(blobs) != self.world or bad:
            raise RuntimeError(f"IPC communicator: {what} failed on rank(s) {bad if bad else '?'} (this rank: {'ok' if ok else 'FAILED'})")
        return [b[1:] for b in blobs]

    def _build(self, ws_bytes: int):
        lib = _native.load()
        if self.ptr is not None:
            torch.cuda.synchronize()
            lib.univst_comm_destroy(self.ptr)
            self.ptr = None
        h = C.c_void_p()
        nb = lib.univst_comm_handle_bytes()
        mine = C.create_string_buffer(nb)
        err = None
        try:
            _native.check(lib.univst_comm_create(self.rank, self.world, int(ws_bytes), C.byref(h)), "comm_create")
            _native.check(lib.univst_comm_export(h, mine), "comm_export")
        except Exception as e:       # noqa: BLE001
            err = e
        try:
            blobs = self._agree(err is None, "create/export", mine.raw)
        except RuntimeError as e:
            if h:
                lib.univst_comm_destroy(h)
            raise e from err
        if any(len(b) != nb for b in blobs):
            lib.univst_comm_destroy(h)
            raise RuntimeError("IPC handle exchange returned a malformed list")
        self.ptr, self.ws_bytes = h, int(ws_bytes)
        try:
            _native.check(lib.univst_comm_connect(h, b"".join(blobs)), "comm_connect")
        except Exception as e:       # noqa: BLE001
            err = e
        try:
            self._agree(err is None, "hipIpcOpenMemHandle / connect")
            # self-test (a device-side collective, entered only when every rank is connected): sum of (rank + 1) * (i + 1) must
            # come out exactly, on every rank — and the ranks agree on THAT outcome too before anyone relies on the communicator
            t = torch.arange(1, 65, device=self.device, dtype=torch.float32) * (self.rank + 1)
            self.all_reduce_sum(t)
            torch.cuda.synchronize()
            want = torch.arange(1, 65, dtype=torch.float32) * (self.world * (self.world + 1) // 2)
            st = lib.univst_comm_status(h)
            self._agree(st == 0 and torch.equal(t.cpu(), want), f"all-reduce self-test (status {st} on rank {self.rank})")
        except RuntimeError as e:
            lib.univst_comm_destroy(h)
            self.ptr = None
            raise e from err

    def ensure_bytes(self, ws_bytes: int):
        """grow the shared workspace (a collective: every rank sees the same latent sizes, hence the same growth sequence)."""
        if ws_bytes > self.ws_bytes:
            self._build(ws_bytes)

    def all_reduce_sum(self, t: torch.Tensor):
        assert t.dtype == torch.float32 and t.is_contiguous()
        _native.check(_native.load().univst_comm_allreduce_f32(self.ptr, t.data_ptr(), t.numel(), _native.stream_ptr()), "comm_allreduce")

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            outs = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(outs, t)
            return outs
        c = t.detach().to("cpu")
        outs = [torch.empty_like(c) for _ in range(self.world)]
        dist.all_gather(outs, c)
        return [o.to(t.device) for o in outs]

    def abort(self):
        pass

    def close(self):
        """Retire this communicator on ALL ranks together (a collective): every rank drains its stream and meets the others at a host barrier
        before anyone unmaps / frees the regions — a peer's kernel may still be writing into this rank's region through its IPC mapping.
        (Device-side waits are bounded — tens of seconds, csrc/comm.hip — so a rank whose sharded forward waits for a dead peer does reach
        this point.)"""
        import torch.distributed as dist
        torch.cuda.synchronize()
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        if self.ptr is not None:
            _native.load().univst_comm_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if self.ptr is not None:
                _native.load().univst_comm_destroy(self.ptr)
        except Exception:
            pass


class EmulatedIpcComm:
    """ONE rank of a `world`-rank job alone on one GPU, through the library's communicator in its emulated mode (``univst_comm_connect_emulated``):
    the production kernels, forked stream and flag waits, with every transfer replaced by a delay of latency + packs-on-the-busiest-link x bytes / rate.
    ``bench.py --emulate-rank r/w --emulate-wire GBPS --comm-emulated``; results are meaningless, only the timing is used."""

    def __init__(self, rank: int, world: int, ws_bytes: int, wire_gbps: float, latency_us: float = 3.0, device=None):
        self.rank, self.world = rank, world
        self.wire_gbps, self.latency_us = float(wire_gbps), float(latency_us)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.ptr, self.ws_bytes = None, 0
        self._build(ws_bytes)

    def _build(self, ws_bytes: int):
        lib = _native.load()
        if self.ptr is not None:
            torch.cuda.synchronize()
            lib.univst_comm_destroy(self.ptr)
            self.ptr = None
        h = C.c_void_p()
        _native.check(lib.univst_comm_create(self.rank, self.world, int(ws_bytes), C.byref(h)), "comm_create")
        _native.check(lib.univst_comm_connect_emulated(h, self.wire_gbps, self.latency_us), "comm_connect_emulated")
        self.ptr, self.ws_bytes = h, int(ws_bytes)

    def ensure_bytes(self, ws_bytes: int):
        if ws_bytes > self.ws_bytes:
            self._build(ws_bytes)

    def all_reduce_sum(self, t: torch.Tensor):
        assert t.dtype == torch.float32 and t.is_contiguous()
        _native.check(_native.load().univst_comm_allreduce_f32(self.ptr, t.data_ptr(), t.numel(), _native.stream_ptr()), "comm_allreduce")

    def all_gather(self, t):
        return [t for _ in range(self.world)]

    def wire_us(self) -> float:
        out = C.c_double(0.0)
        _native.check(_native.load().univst_comm_query(self.ptr, b"emu_wire_us", C.byref(out)), "comm_query")
        return out.value

    def close(self):
        if self.ptr is not None:
            torch.cuda.synchronize()
            _native.load().univst_comm_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TorchDistComm:
    """RCCL (or gloo on CPU) through torch.distributed; all ops are enqueued against the current stream."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_reduce_sum(self, t: torch.Tensor):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t, group=self.group)
        return outs

    def halo_and_broadcast(self, send_last, first, recv_prev, recv_first):
        """rank 0 broadcasts `first` (others receive into `recv_first`); r -> r+1 halo of `send_last` into `recv_prev`."""
        dist, r, w = self.dist, self.rank, self.world
        dist.broadcast(first if r == 0 else recv_first, src=0, group=self.group)
        ops = []
        if r < w - 1:
            ops.append(dist.P2POp(dist.isend, send_last, r + 1, self.group))
        if r > 0:
            ops.append(dist.P2POp(dist.irecv, recv_prev, r - 1, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()


class HostStagedDistComm(TorchDistComm):
    """torch.distributed backend without device support for some ops (gloo): stage every payload through pinned host
    memory.  Used to exercise the multi-process path on boxes where the ranks cannot each own a GPU (bring-up / CI);
    production uses TorchDistComm over RCCL."""

    def _cpu(self, t):
        return t.detach().to("cpu")

    def all_reduce_sum(self, t):
        c = self._cpu(t)
        self.dist.all_reduce(c, op=self.dist.ReduceOp.SUM, group=self.group)
        t.copy_(c)

    def all_gather(self, t):
        c = self._cpu(t)
        outs = [torch.empty_like(c) for _ in range(self.world)]
        self.dist.all_gather(outs, c, group=self.group)
        return [o.to(t.device) for o in outs]

    def halo_and_broadcast(self, send_last, first, recv_prev, recv_first):
        dist, r, w = self.dist, self.rank, self.world
        c = self._cpu(first if r == 0 else recv_first)
        dist.broadcast(c, src=0, group=self.group)
        if r > 0:
            recv_first.copy_(c)
        reqs = []
        if r < w - 1:
            reqs.append(dist.isend(self._cpu(send_last), r + 1, group=self.group))
        rp = None
        if r > 0:
            rp = torch.empty(recv_prev.shape, dtype=recv_prev.dtype)
            reqs.append(dist.irecv(rp, r - 1, group=self.group))
        for q in reqs:
            q.wait()
        if rp is not None:
            recv_prev.copy_(rp)


class NullComm:
    """diagnostic stand-in (``bench.py --emulate-rank r/w``): every collective returns at once, so ONE process on one GPU
    runs exactly the kernels, pack/unpack copies and host callbacks of rank r of a w-GPU job — everything except the wire.
    Results are meaningless (nothing is exchanged); only the timing is used."""

    def __init__(self, rank, world, wire_gbps=None, latency_us=3.0, kv_in_library=False):
        """kv_in_library: the K/V exchanges' wire time is issued by the library on its forked stream (``emu_wire_gbps`` option of the UNet handle, round 6);
        this object then only models the all-reduces' flag round trips.
        wire_gbps (``bench.py --emulate-wire``): per-direction rate of ONE xGMI link; every exchange then occupies the stream for the time its
        slowest transfer would take, in place — the serial issue order of csrc/comm.hip (pack -> multicast -> raise -> wait on the compute stream).
        The first-frame pack reaches every rank over its own link from rank 0 and the halo pack over the link from rank - 1: one pack per link,
        except on rank 1, whose single link from rank 0 carries both; each GroupNorm all-reduce costs one flag round trip (latency_us)."""
        self.rank, self.world = rank, world
        self.wire_gbps, self.latency_us, self.kv_in_library = wire_gbps, latency_us, kv_in_library
        self.wire_us = 0.0            # modelled wire time accumulated since the last reset (bench.py reports it per step)

    def _delay(self, us):
        if self.wire_gbps:
            self.wire_us += us
            _native.delay_us(us)

    def all_reduce_sum(self, t):
        self._delay(self.latency_us)

    def all_gather(self, t):
        return [t for _ in range(self.world)]

    def halo_and_broadcast(self, send_last, first, recv_prev, recv_first):
        if self.wire_gbps and self.world > 1 and not self.kv_in_library:
            packs = 2 if self.rank == 1 else 1
            self._delay(self.latency_us + packs * send_last.numel() * send_last.element_size() / (self.wire_gbps * 1e3))


class ThreadLoopbackComm:
    """`world` host threads of ONE process play the ranks (each with its own native UNet handle and its own HIP stream
    on the same GPU).  Collectives are rendezvous on a threading.Barrier plus device copies — the same call sites and
    the same buffers as the RCCL path, so the exchange schedule is validated on a 1-GPU box."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots: List[Optional[dict]] = [None] * world

    def __init__(self, shared: "ThreadLoopbackComm.Shared", rank: int):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def _publish(self, **tensors):
        torch.cuda.current_stream().synchronize()        # my data is complete before the others read it
        self.sh.slots[self.rank] = tensors
        self.sh.barrier.wait()

    def _done(self):
        torch.cuda.current_stream().synchronize()        # my reads are complete before the others overwrite
        self.sh.barrier.wait()

    def all_reduce_sum(self, t):
        self._publish(t=t)
        total = sum(self.sh.slots[r]["t"].clone() for r in range(self.world))
        self._done()
        t.copy_(total)

    def all_gather(self, t):
        self._publish(t=t)
        outs = [self.sh.slots[r]["t"].clone() for r in range(self.world)]
        self._done()
        return outs

    def halo_and_broadcast(self, send_last, first, recv_prev, recv_first):
        self._publish(send_last=send_last, first=first)
        if self.rank > 0:
            recv_first.copy_(self.sh.slots[0]["first"])
            recv_prev.copy_(self.sh.slots[self.rank - 1]["send_last"])
        self._done()
