"""UNetModel of the reference's improved_diffusion/unet.py:278-477 as a HIP-backed module.  The parameter containers
keep the reference's names (time_embed, input_blocks.<i>.<j>.{in_layers,emb_layers,out_layers,skip_connection,norm,qkv,
proj_out,op}, middle_block, output_blocks, out) so that checkpoints load unchanged; the forward pass runs in
libdmad_hip.so (dmad_unet_eps): convs / linears on the fp32 matrix cores over NHWC maps, GroupNorm32 + SiLU +
scale-shift, 4-head attention and nearest upsampling as small HIP kernels.  Inference only; all rows of a batch carry
the same timestep (what p_sample_loop passes)."""
import torch
import torch.nn as nn


def _gn(ch):
    return nn.GroupNorm(32, ch)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, out_channels):
        super().__init__()
        self.in_layers = nn.Sequential(_gn(channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * out_channels))
        self.out_layers = nn.Sequential(_gn(out_channels), nn.SiLU(), nn.Dropout(p=0.0), nn.Conv2d(out_channels, out_channels, 3, padding=1))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)


class AttentionBlock(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.norm = _gn(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = nn.Conv1d(channels, channels, 1)


class Downsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.op = nn.Conv2d(channels, channels, 3, stride=2, padding=1)


class Upsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)


class UNetModel(nn.Module):

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, num_heads=1,
                 num_heads_upsample=-1, use_scale_shift_norm=False):
        super().__init__()
        from dmad_hip import synth
        cfg = dict(in_channels=in_channels, model_channels=model_channels, out_channels=out_channels, num_res_blocks=num_res_blocks,
                   attention_resolutions=tuple(attention_resolutions), channel_mult=tuple(channel_mult), num_heads=num_heads,
                   use_scale_shift_norm=use_scale_shift_norm)
        if cfg != synth.UNET_CONFIG or num_classes is not None or dims != 2 or not conv_resample or num_heads_upsample not in (-1, num_heads):
            raise NotImplementedError('the HIP engine builds the UNet of the reference wrapper only: %r' % (synth.UNET_CONFIG,))
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, tuple(attention_resolutions), tuple(channel_mult)
        self.num_classes, self.num_heads = None, num_heads
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        _, inp, mid, outp = synth.unet_layout(cfg)

        def build(blk):
            mods = []
            for _, kind, cin, cout in blk:
                mods.append({'conv_in': lambda: nn.Conv2d(cin, cout, 3, padding=1), 'res': lambda: ResBlock(cin, ted, cout),
                             'attn': lambda: AttentionBlock(cin), 'down': lambda: Downsample(cin), 'up': lambda: Upsample(cin)}[kind]())
            return nn.Sequential(*mods)
        self.input_blocks = nn.ModuleList([build(b) for b in inp])
        self.middle_block = build(mid)
        self.output_blocks = nn.ModuleList([build(b) for b in outp])
        self.out = nn.Sequential(_gn(model_channels), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))

    # -- HIP engine binding ---------------------------------------------------------------------
    def bind_engine(self, engine=None):
        from dmad_hip import engine as _eng
        eng = engine or _eng.get_engine()
        eng.bind('unet', self.state_dict(), eng.load_unet)
        self.__dict__['engine'] = eng
        return self

    def forward(self, x, timesteps, y=None):
        assert y is None, 'must specify y if and only if the model is class-conditional'
        if self.training:
            raise NotImplementedError('the HIP UNet is inference-only: call .eval() first')
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError('the HIP UNet is inference-only (no autograd)')
        steps = torch.as_tensor(timesteps).detach().reshape(-1).float().cpu()
        t = float(steps[0])
        if not bool((steps == t).all()) or t != int(t):
            raise NotImplementedError('per-row / fractional timesteps are not supported by the HIP engine')
        if 'engine' not in self.__dict__:
            self.bind_engine()
        return self.__dict__['engine'].unet_eps(x, int(t)).unsqueeze(1)
