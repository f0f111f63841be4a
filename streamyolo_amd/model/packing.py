"""Weight repacking for the MFMA kernels (parameters stay ordinary OIHW fp32
nn.Parameters owned by torch; the kernels borrow K-contiguous packed copies)."""
import torch

from ..ops import TORCH_DTYPE


def pack_conv_weight(w, dtype_code, transpose=False, pad_cin_to=None, pad_cout_to=None):
    """OIHW fp32 -> [Cout][kh*kw][Cin] (forward / wgrad layout) or, with transpose=True,
    [Cin][kh*kw][Cout] (the data-gradient's "weights"), flattened to 2-D, in the compute dtype."""
    co, ci, kh, kw = w.shape
    w = w.detach()
    if pad_cin_to is not None and pad_cin_to > ci:
        w = torch.cat([w, w.new_zeros(co, pad_cin_to - ci, kh, kw)], 1)
        ci = pad_cin_to
    if pad_cout_to is not None and pad_cout_to > co:
        w = torch.cat([w, w.new_zeros(pad_cout_to - co, ci, kh, kw)], 0)
        co = pad_cout_to
    if transpose:
        p = w.permute(1, 2, 3, 0).reshape(ci, kh * kw * co)
    else:
        p = w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
    return p.to(TORCH_DTYPE[dtype_code]).contiguous()


def fold_bn(gamma, beta, mean, var, eps):
    """Eval-mode BatchNorm as a per-channel affine on the fp32 accumulator (no precision is lost
    by folding into low-precision weights): scale = g/sqrt(v+eps), shift = b - m*scale."""
    scale = gamma.detach().float() / torch.sqrt(var.detach().float() + eps)
    shift = beta.detach().float() - mean.detach().float() * scale
    return scale.contiguous(), shift.contiguous()


def pack_conv_weight_frag(packed, ksize):
    """[Cout, k*k*Cin] K-contiguous packed weights -> MFMA-fragment order along the FAST kernel's K
    traversal (channel slab outer, taps inner): [Cout/32][Cin/BK][taps][g][half][32 rows][EPC], i.e. one
    wave instruction of the SY_TILE_WR kernels reads 1 KiB contiguous.  Returns None when Cin is not a
    whole number of 64-byte slabs (those layers use the generic loader)."""
    cout, K = packed.shape
    taps = ksize * ksize
    cin = K // taps
    epc = 16 // packed.element_size()
    bk = 4 * epc
    if cin % bk != 0:
        return None
    pad = (-cout) % 32
    if pad:
        packed = torch.cat([packed, packed.new_zeros(pad, K)], 0)
    ct = packed.shape[0] // 32
    v = packed.view(ct, 32, taps, cin // bk, 2, 2, epc)          # [ct, r, t, c, g, half, e]
    return v.permute(0, 3, 2, 4, 5, 1, 6).contiguous().view(-1)   # [ct, c, t, g, half, r, e]
