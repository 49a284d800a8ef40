"""Column-sharded int8 linear layers over the GPUs of one node (one process per GPU).

Every output channel n of a quantized Linear owns its row of W, its scale, its bias and its
column of the output (the reference quantizes row by row: quant/Weight_Only/Qop.py:81-87), so
rank r of g keeps rows [lo_r, hi_r) of W[N,K] and computes out[:, lo_r:hi_r] with no
communication; the quantizer needs none either.  The single exchange is an all-gather of the
[M, N/g] output blocks (RCCL over xGMI through torch.distributed's "nccl" backend).  The
reference has no multi-GPU code; the gathered result is bit-identical to the single-GPU one
because no accumulation order changes.

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


class ColumnShardedLinear:
    """This rank's slice of a weight-only int8 Linear, plus the gather that rebuilds [M, N].

    weight_q [n_local, K] int8, scale [n_local] fp32, bias [n_local] fp32 or None are the slices
    for rows shard_rows(out_features, world, rank).  `gemm(x, w, bias=, alpha=, beta=, out_dtype=)`
    is the product + epilogue; it defaults to the HIP int8 GEMM.
    """

    def __init__(self, weight_q, scale, bias, out_features, group=None, out_dtype=torch.float32, gemm=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.out_features = out_features
        self.lo, self.hi = shard_rows(out_features, self.world, self.rank)
        if weight_q.shape[0] != self.hi - self.lo:
            raise RuntimeError(f"rank {self.rank} owns rows [{self.lo},{self.hi}) but was given {weight_q.shape[0]}")
        self.weight, self.scale, self.bias = weight_q, scale, bias
        self.out_dtype = out_dtype
        self.n_max = -(-out_features // self.world)
        self._gemm = gemm or _native.gemm_i8
        self._custom_gemm = gemm is not None

    @classmethod
    def from_float(cls, weight, bias=None, group=None, out_dtype=torch.float32, gemm=None, quantize=None):
        """Quantize this rank's rows of a full fp weight [N, K] (each rank may also pass only
        its own rows with `weight.shape[0] == hi - lo`, which is how large models are loaded)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        n = weight.shape[0]
        lo, hi = shard_rows(n, world, rank)
        q, scale = (quantize or _quantize_hip)(weight[lo:hi])
        b = None if bias is None else bias[lo:hi].to(torch.float32).contiguous()
        return cls(q, scale, b, n, group=group, out_dtype=out_dtype, gemm=gemm)

    def local_forward(self, x):
        """out[:, lo:hi] for activations x [M, K] replicated on every rank: int8 (int8 product) or floating point
        (the weight-only product on the int8 weight as stored)."""
        if self._custom_gemm or x.dtype == torch.int8:
            return self._gemm(x, self.weight, bias=self.bias, alpha=self.scale, beta=0.0, out_dtype=self.out_dtype)
        return _native.gemm_w8(x, self.weight, bias=self.bias, alpha=self.scale, beta=0.0, out_dtype=self.out_dtype)

    def gather_async(self, y_local):
        """Start the all-gather of the ranks' column blocks.  Returns (buffer [g, M, n_max], work)."""
        M = y_local.shape[0]
        if self.world == 1:
            return y_local.unsqueeze(0), None
        if y_local.shape[1] != self.n_max:  # ragged split: pad the narrower blocks
            padded = y_local.new_zeros((M, self.n_max))
            padded[:, :y_local.shape[1]] = y_local
            y_local = padded
        buf = torch.empty((self.world, M, self.n_max), dtype=y_local.dtype, device=y_local.device)
        # the output is the ranks' blocks concatenated along dim 0: [g*M, n_max]
        work = dist.all_gather_into_tensor(buf.view(self.world * M, self.n_max), y_local.contiguous(),
                                           group=self.group, async_op=True)
        return buf, work

    def assemble(self, buf, work=None):
        """[g, M, n_max] blocks -> the [M, N] output a single GPU would have produced."""
        if work is not None:
            work.wait()
        g, M, _ = buf.shape
        if g == 1:
            return buf[0]
        if self.out_features % g == 0:
            return buf.permute(1, 0, 2).reshape(M, self.out_features)
        cols = [buf[r, :, :shard_rows(self.out_features, g, r)[1] - shard_rows(self.out_features, g, r)[0]]
                for r in range(g)]
        return torch.cat(cols, dim=1)

    def forward(self, x):
        buf, work = self.gather_async(self.local_forward(x))
        return self.assemble(buf, work)

    __call__ = forward


def run_stack(layers, inputs, assemble=True):
    """Apply independent sharded layers, overlapping each all-gather with the next GEMM.

    layers: list of ColumnShardedLinear; inputs: one activation tensor per layer.
    Returns the list of [M, N] outputs (or the raw (buffer, work) pairs with assemble=False).
    """
    pending = []
    for layer, x in zip(layers, inputs):
        pending.append((layer,) + layer.gather_async(layer.local_forward(x)))
    if not assemble:
        for _, _, work in pending:
            if work is not None:
                work.wait()
        return [buf for _, buf, _ in pending]
    return [layer.assemble(buf, work) for layer, buf, work in pending]
