"""Column-sharded int8 linear layers over the GPUs of one node (one process per GPU).

Every output channel n of a quantized Linear owns its row of W, its scale, its bias and its
column of the output (the reference quantizes row by row: quant/Weight_Only/Qop.py:81-87), so
rank r of g keeps rows [lo_r, hi_r) of W[N,K] and computes out[:, lo_r:hi_r] with no
communication; the quantizer needs none either.  The single exchange is an all-gather of the
[M, N/g] output blocks (RCCL over xGMI through torch.distributed's "nccl" backend).  The
reference has no multi-GPU code; the gathered fp32 result is bit-identical to the single-GPU one
because no accumulation order changes.

Data movement.  The GEMM writes its block straight into this rank's slot of the gather buffer and
the all-gather runs in place on that buffer (input = the rank's slot of the output), so a layer
costs one kernel, one collective, no padding pass and no staging copy.  Two placements:

* blocks (default): the gathered result is the list of the ranks' column blocks, `blocks()`: block r
  is the dense [M, n_r] matrix out[:, lo_r:hi_r].  A single row-major [M, N] matrix needs the blocks
  interleaved row by row, which a collective over contiguous chunks cannot place: `to_dense()` pays
  one pass for callers that want it.
* transposed=True: every rank computes the TRANSPOSED block out^T[lo_r:hi_r, :] (the GEMM with the
  operands' roles swapped, per-row scale and bias: miq_gemm_i8_ex, MIQ_OUT_NM).  The ranks' blocks are
  then consecutive rows of one [N, M] matrix, so the gathered buffer IS the dense transposed output and
  `to_dense()` is a view of it, `buf.t()` with shape [M, N], no pass at all (when N divides by the
  world size; a ragged split closes the gaps between the slots with one copy).

`forward()` returns tensors the caller owns (a fresh buffer per call).  `gather_async(x, slot=k)`
reuses one cached buffer per (rows, slot): its result is valid until the next call with the same
rows and slot, which is what a stack of layers whose gathers overlap wants (`run_stack`, bench.py).

What travels.  `gather_dtype` chooses the bytes on the links: the output dtype itself (fp32:
bit-identical to one GPU), bfloat16 (half the bytes; the epilogue rounds once), or int8 (a
quarter): the GEMM's epilogue requantises with a static activation scale,
x_q = clamp(rint(y / act_scale)), the per-tensor form of Qop.quantize
(quant/Weight_Only/Qop.py:154-157), which is what a following int8 layer consumes.

Layers of a stack that read the same activations are independent, so the all-gather of layer i
runs on the communication stream while the GEMM of layer i+1 runs on the compute stream.
"""
import torch
import torch.distributed as dist

from . import _native


def shard_rows(n, world, rank):
    """[lo, hi) of the output channels owned by `rank`: contiguous, sizes differ by at most 1."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _quantize_hip(w):
    q, scale, _ = _native.quantize_rows_sym_i8(w.contiguous())
    return q, scale


class Gathered:
    """The ranks' blocks of one layer's output, in the gather buffer [g, M * n_max] (transposed: [g * n_max, M])."""

    def __init__(self, buf, rows, out_features, work=None, transposed=False):
        self.buf, self.rows, self.out_features, self.work, self.transposed = buf, rows, out_features, work, transposed

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self

    def blocks(self):
        """[out[:, lo_r:hi_r] for r in ranks] as [M, n_r] views of the buffer, no copy (dense; transposed placement: the
        transposes of the buffer's row blocks, i.e. column-major views)."""
        self.wait()
        g, M = self.buf.shape[0], self.rows
        views = []
        if self.transposed:
            n_max = -(-self.out_features // g)
            flat = self.buf.view(g, n_max, M)
            for r in range(g):
                lo, hi = shard_rows(self.out_features, g, r)
                views.append(flat[r, :hi - lo].t())
            return views
        for r in range(g):
            lo, hi = shard_rows(self.out_features, g, r)
            views.append(self.buf[r, :M * (hi - lo)].view(M, hi - lo))
        return views

    def to_dense(self):
        """The [M, N] matrix a single GPU would have produced.  Blocks placement: one pass over the output for more
        than one rank.  Transposed placement: a view of the gathered buffer ([N, M] storage, returned as its
        transpose), no pass, when N divides by the world size."""
        self.wait()
        g, M, N = self.buf.shape[0], self.rows, self.out_features
        if self.transposed:
            n_max = -(-N // g)
            if n_max * g == N:
                return self.buf.view(N, M).t()
            flat = self.buf.view(g, n_max, M)
            parts = [flat[r, :shard_rows(N, g, r)[1] - shard_rows(N, g, r)[0]] for r in range(g)]
            return torch.cat(parts, dim=0).t()
        b = self.blocks()
        return b[0] if len(b) == 1 else torch.cat(b, dim=1)


class ColumnShardedLinear:
    """This rank's slice of a weight-only int8 Linear, plus the gather that rebuilds the output.

    weight_q [n_local, K] int8, scale [n_local] fp32, bias [n_local] fp32 or None are the slices
    for rows shard_rows(out_features, world, rank).  `gemm(x, w, bias=, alpha=, beta=, out_dtype=)`
    is the product + epilogue; it defaults to the HIP kernels (and then writes in place).
    gather_dtype: None (= out_dtype), torch.bfloat16, or torch.int8 (needs act_scale > 0).
    transposed: every rank computes and gathers out^T (see the module docstring); int8 activations only.
    """

    def __init__(self, weight_q, scale, bias, out_features, group=None, out_dtype=torch.float32, gemm=None,
                 gather_dtype=None, act_scale=None, transposed=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.out_features = out_features
        self.lo, self.hi = shard_rows(out_features, self.world, self.rank)
        if weight_q.shape[0] != self.hi - self.lo:
            raise RuntimeError(f"rank {self.rank} owns rows [{self.lo},{self.hi}) but was given {weight_q.shape[0]}")
        self.weight, self.scale, self.bias = weight_q, scale, bias
        self.gather_dtype = gather_dtype or out_dtype
        if self.gather_dtype == torch.int8:
            if act_scale is None or not act_scale > 0:
                raise RuntimeError("an int8 gather needs the activation scale (act_scale > 0)")
            self.act_scale = float(act_scale)
            out_dtype = torch.int8             # requantised by the GEMM's epilogue, straight into the slot
        elif self.gather_dtype == torch.bfloat16:
            out_dtype = torch.bfloat16         # the epilogue rounds once, straight into the slot
        elif self.gather_dtype != out_dtype:
            raise RuntimeError(f"gather_dtype must be None, bfloat16 or int8 (got {gather_dtype})")
        self.out_dtype = out_dtype
        self.transposed = bool(transposed)
        self.n_max = -(-out_features // self.world)
        self._gemm = gemm
        self._bufs = {}

    @classmethod
    def from_float(cls, weight, bias=None, group=None, out_dtype=torch.float32, gemm=None, quantize=None,
                   out_features=None, **kw):
        """Quantize this rank's rows of an fp weight.  `weight` is the full [N, K] matrix (this rank's
        rows are sliced out of it and of `bias`), or, with `out_features=N` given, only this rank's
        rows [hi - lo, K] and bias entries: how large models are loaded, no rank ever holding N x K floats."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        if out_features is None:
            n = weight.shape[0]
            lo, hi = shard_rows(n, world, rank)
            weight = weight[lo:hi]
            bias = None if bias is None else bias[lo:hi]
        else:
            n = out_features
            lo, hi = shard_rows(n, world, rank)
            if weight.shape[0] != hi - lo or (bias is not None and bias.shape[0] != hi - lo):
                raise RuntimeError(f"rank {rank} owns rows [{lo},{hi}) of {n}: expected {hi - lo} rows, got {weight.shape[0]}")
        q, scale = (quantize or _quantize_hip)(weight)
        b = None if bias is None else bias.to(torch.float32).contiguous()
        return cls(q, scale, b, n, group=group, out_dtype=out_dtype, gemm=gemm, **kw)

    # ---- buffers ----------------------------------------------------------------------------
    def _gather_buffer(self, M, device, slot=None):
        """[g, M * n_max] of gather_dtype.  slot = None: a fresh buffer the caller owns; slot = k: one buffer per
        (M, device, k), reused call after call (`slot` tells apart the buffers of layers whose gathers are in flight at
        the same time)."""
        if slot is None:
            return torch.empty((self.world, M * self.n_max), dtype=self.gather_dtype, device=device)
        key = (M, device, slot)
        buf = self._bufs.get(key)
        if buf is None:
            buf = self._bufs[key] = torch.empty((self.world, M * self.n_max), dtype=self.gather_dtype, device=device)
        return buf

    def _product(self, x, out):
        """out <- epilogue(x . W_local^T): [M, n_local], or its transpose [n_local, M] for the transposed placement
        (dense, of self.out_dtype; int8: requantised with the static activation scale)."""
        if self._gemm is not None:     # a stand-in product (CPU tests): no in-place form, no fused requantisation
            dt = torch.float32 if self.out_dtype == torch.int8 else self.out_dtype
            y = self._gemm(x, self.weight, bias=self.bias, alpha=self.scale, beta=0.0, out_dtype=dt)
            if self.out_dtype == torch.int8:
                y = torch.clamp(torch.round(y / torch.tensor(self.act_scale, dtype=torch.float32)), -128, 127).to(torch.int8)
            out.copy_(y.t() if self.transposed else y)
        elif x.dtype == torch.int8:
            _native.gemm_i8(x, self.weight, bias=self.bias, alpha=self.scale, beta=0.0, out_dtype=self.out_dtype, out=out,
                            transpose_out=self.transposed,
                            requant_scale=self.act_scale if self.out_dtype == torch.int8 else 1.0)
        else:
            if self.transposed or self.out_dtype == torch.int8:
                raise RuntimeError("floating-point activations support the blocks placement with fp32 / bf16 gathers only")
            _native.gemm_w8(x, self.weight, bias=self.bias, alpha=self.scale, beta=0.0, out_dtype=self.out_dtype, out=out)
        return out

    def local_forward(self, x, out=None):
        """out[:, lo:hi] for activations x [M, K] replicated on every rank: int8 (int8 product) or floating
        point (the weight-only product on the int8 weight as stored).  Transposed placement: out^T[lo:hi, :]."""
        M, n = x.shape[0], self.hi - self.lo
        if out is None:
            out = torch.empty((n, M) if self.transposed else (M, n), dtype=self.out_dtype, device=x.device)
        return self._product(x, out)

    def compute_block(self, x, slot=0):
        """This rank's block computed straight into its slot of the gather buffer (no collective yet)."""
        M, n = x.shape[0], self.hi - self.lo
        buf = self._gather_buffer(M, x.device, slot)
        mine = buf[self.rank, :M * n].view((n, M) if self.transposed else (M, n))
        self._product(x, mine)
        return Gathered(buf, M, self.out_features, None, self.transposed)

    def start_gather(self, got):
        """The in-place all-gather of a computed block (asynchronous: got.wait() before reading other ranks' blocks)."""
        if self.world > 1:
            if got.buf.is_cuda and dist.get_backend(self.group) != "nccl":
                # A host backend (gloo: rehearsals with every rank on one GPU) is handed host tensors: its own
                # staging of device tensors stalled intermittently on a shared GPU (measured).  Synchronous.
                send = got.buf[self.rank].cpu()
                recv = torch.empty(got.buf.shape, dtype=got.buf.dtype)
                dist.all_gather_into_tensor(recv.view(-1), send, group=self.group)
                got.buf.copy_(recv)
            else:
                got.work = dist.all_gather_into_tensor(got.buf.view(-1), got.buf[self.rank], group=self.group, async_op=True)
        return got

    def gather_async(self, x, slot=0):
        """compute_block + start_gather.  Returns a Gathered (wait() / blocks() / to_dense())."""
        return self.start_gather(self.compute_block(x, slot))

    def forward(self, x, dense=True):
        """The layer's output on every rank: [M, N] (dense=True) or the list of the ranks' column blocks.  The result
        lives in a buffer of its own: a later call does not overwrite it."""
        got = self.gather_async(x, slot=None)
        return got.to_dense() if dense else got.blocks()

    __call__ = forward


def run_stack(layers, inputs, dense=True):
    """Apply independent sharded layers, overlapping each all-gather with the next GEMM.

    layers: list of ColumnShardedLinear; inputs: one activation tensor per layer.
    Returns the list of [M, N] outputs (dense=True) or of block lists, each in a buffer the caller owns.
    """
    pending = [layer.gather_async(x, slot=None) for layer, x in zip(layers, inputs)]
    return [p.to_dense() if dense else p.blocks() for p in pending]
