"""Golden vectors for the relative-position attention core, produced by the HuggingFace XLNet code that IS installed in the build
container (transformers 5.15: models/xlnet/modeling_xlnet.py).  Transformer-XL itself was dropped from transformers, but XLNet
inherits its attention core unchanged: `rel_attn_core` = (q + r_w_bias).k  +  rel_shift((q + r_r_bias).R)  -> scale -> mask ->
softmax -> .v, with R = r_net(sinusoid [sin || cos] of the clamped relative positions).  With segment terms off and the
Transformer-XL `same_length` mask passed in explicitly this is exactly upstream RelPartialLearnableMultiHeadAttn's core
(SURVEY A.2-A.4), so these vectors pin oracle/relattn_ref.py, oracle/transfoxl_ref.py's rel-shift / sinusoid and the HIP kernel
on an external implementation.

    python tests/golden/make_xlnet_relattn_goldens.py      # writes tests/golden/xlnet_relattn_core.pt   (inputs + outputs only)
"""
import os

import torch
from transformers import XLNetConfig
from transformers.models.xlnet.modeling_xlnet import XLNetLayer, XLNetModel, XLNetRelativeAttention

HERE = os.path.dirname(os.path.abspath(__file__))


def case(seed, qlen, mlen, H, dh, B, clamp_len, keep_intermediates=True):
    torch.manual_seed(seed)
    d_model = H * dh
    klen = qlen + mlen
    cfg = XLNetConfig(d_model=d_model, n_head=H, d_head=dh, d_inner=4 * d_model, n_layer=1, dropout=0.0, vocab_size=32,
                      attn_type='uni', bi_data=False, clamp_len=clamp_len, mem_len=mlen, same_length=False)
    attn = XLNetRelativeAttention(cfg).eval()
    with torch.no_grad():
        attn.r_w_bias.copy_(torch.randn(H, dh) * 0.3)
        attn.r_r_bias.copy_(torch.randn(H, dh) * 0.3)
    q = torch.randn(qlen, B, H, dh)
    k = torch.randn(klen, B, H, dh)
    v = torch.randn(klen, B, H, dh)
    r_weight = torch.randn(d_model, H, dh) / d_model ** 0.5          # XLNet's `r` projection = Transformer-XL's r_net
    # relative positions klen .. 0 ('uni'), clamped, [sin || cos] -- XLNetModel.relative_positional_encoding, verbatim call
    model = XLNetModel(cfg).eval()
    pos_emb = model.relative_positional_encoding(qlen, klen, bsz=B)                 # (klen + 1, B, d_model)
    k_head_r = torch.einsum('ibh,hnd->ibnd', pos_emb, r_weight)                     # (klen + 1, B, H, dh)
    # Transformer-XL same_length mask for mlen == mem_len: query i sees keys j with i < j <= i + mlen  (1.0 = masked)
    i = torch.arange(qlen)[:, None]
    j = torch.arange(klen)[None, :]
    mask = ((j > i + mlen) | (j <= i)).float()[:, :, None, None]
    with torch.no_grad():
        attn_vec, attn_prob = attn.rel_attn_core(q, k, v, k_head_r, seg_mat=None, attn_mask=mask, output_attentions=True)
        bd_raw = torch.einsum('ibnd,jbnd->bnij', q + attn.r_r_bias, k_head_r)
        bd_shifted = attn.rel_shift_bnij(bd_raw, klen=klen)
    out = dict(qlen=qlen, mlen=mlen, H=H, dh=dh, B=B, clamp_len=clamp_len,
               q=q, k=k, v=v, r_w_bias=attn.r_w_bias.detach().clone(), r_r_bias=attn.r_r_bias.detach().clone(),
               r_weight=r_weight, pos_emb=pos_emb[:, 0].clone(), mask=mask[:, :, 0, 0].clone(), attn_vec=attn_vec)
    if keep_intermediates:
        out.update(bd_shifted=bd_shifted, attn_prob=attn_prob)
    return out


def layer_case(seed, qlen, mlen, H, dh, B, clamp_len):
    """A whole XLNetLayer (relative attention with its q/k/v/r/o projections over cat(mems, h), post-LayerNorm residual,
    relu feed-forward with post-LayerNorm) = one upstream Transformer-XL DecoderLayer when segment terms are off."""
    torch.manual_seed(seed)
    d_model = H * dh
    klen = qlen + mlen
    cfg = XLNetConfig(d_model=d_model, n_head=H, d_head=dh, d_inner=4 * d_model, n_layer=1, dropout=0.0, vocab_size=32,
                      attn_type='uni', bi_data=False, clamp_len=clamp_len, mem_len=mlen, same_length=False,
                      ff_activation='relu', layer_norm_eps=1e-5)
    layer = XLNetLayer(cfg).eval()
    with torch.no_grad():
        for name, prm in layer.named_parameters():
            if 'layer_norm.weight' in name:
                prm.copy_(1.0 + 0.1 * torch.randn_like(prm))
            elif prm.dim() == 1:
                prm.copy_(0.1 * torch.randn_like(prm))
            else:
                prm.copy_(torch.randn_like(prm) / d_model ** 0.5)
    h = torch.randn(qlen, B, d_model)
    mems = torch.randn(mlen, B, d_model)
    pos_emb = XLNetModel(cfg).eval().relative_positional_encoding(qlen, klen, bsz=B)
    i = torch.arange(qlen)[:, None]
    j = torch.arange(klen)[None, :]
    mask = ((j > i + mlen) | (j <= i)).float()[:, :, None, None]
    with torch.no_grad():
        out = layer(h, None, mask, None, pos_emb, None, mems=mems)[0]
    return dict(qlen=qlen, mlen=mlen, H=H, dh=dh, B=B, clamp_len=clamp_len, h=h, mems=mems, mask=mask[:, :, 0, 0].clone(),
                params={k: v.detach().clone() for k, v in layer.state_dict().items() if 'seg_embed' not in k and 'r_s_bias' not in k},
                out=out)


if __name__ == '__main__':
    cases = [case(1, qlen=24, mlen=24, H=2, dh=16, B=2, clamp_len=-1),
             case(2, qlen=40, mlen=40, H=3, dh=32, B=1, clamp_len=25),       # clamp_len bites: distances > 25 share a row
             case(3, qlen=64, mlen=64, H=2, dh=64, B=2, clamp_len=-1, keep_intermediates=False)]
    out = os.path.join(HERE, 'xlnet_relattn_core.pt')
    torch.save(cases, out)
    print('wrote', out, os.path.getsize(out) // 1024, 'KiB')
    layers = [layer_case(11, qlen=16, mlen=16, H=2, dh=16, B=2, clamp_len=-1),
              layer_case(12, qlen=24, mlen=24, H=4, dh=16, B=1, clamp_len=20)]
    out = os.path.join(HERE, 'xlnet_layer.pt')
    torch.save(layers, out)
    print('wrote', out, os.path.getsize(out) // 1024, 'KiB')
