#!/usr/bin/env python3
"""Evidence for the "power-limited" reading of the layer kernel: board power, the driver's sclk and the
clock the chip holds INSIDE the kernel, sampled while wn_layer_p runs back to back.

  1. a sampler thread reads the card's hwmon power (power1_average / power1_input, uW), its cap (power1_cap) and the current
     sclk (hwmon freq1_input or the starred line of pp_dpm_sclk) every 50 ms — sysfs reads, no tool in between; `rocm-smi
     --showpower --showclocks` is tried once as a cross-check when sysfs is not readable;
  2. the main thread launches the layer kernel continuously for SECONDS (engine batch B, layer 5: dilation 32) on random data
     and on ZERO data (same binary, same cycles: MI355X_MICROARCH.md, DVFS give-back item 1) and reports ms per launch;
  3. the diagnostic (stamped) build of the same kernel then reports the in-kernel clock = shader cycles / 100 MHz ticks
     (give-back item 6) after the chip has been under that load for SECONDS.
  4. MIX=1: the energy of a tile split by operand — the same launches with ONLY the weights of the timed layer zeroed (random
     activations: what the activation path + gate + stores cost) and with everything BUT that layer's weights zeroed (zero
     activations, random weights: what the weight fill + A-fragment path costs); energy per launch = board power x time.
Writes gpurun_out/power_trace.json.       B=256 SECONDS=6 HALF=f16 python tools/gpu_power_trace.py
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import _lib  # noqa: E402
if os.environ.get('DMAD_LIB'):                   # A/B of two builds on one box (tools/layer_variants.sh)
    _lib.LIB_PATH = os.environ['DMAD_LIB']
from dmad_hip import engine as E, synth  # noqa: E402

B = int(os.environ.get('B', 256))
SECONDS = float(os.environ.get('SECONDS', 6))
HALF = os.environ.get('HALF', 'f16')
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)


def first_readable(paths):
    for p in paths:
        try:
            with open(p) as f:
                f.read()
            return p
        except Exception:
            continue
    return None


def pci_slot_of_hip_device():
    """'0000:c5:00.0'-style PCI address of HIP device 0 (torch's device properties), or None."""
    try:
        p = torch.cuda.get_device_properties(0)
        return '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def sysfs_sources():
    """hwmon files of the card whose PCI address is HIP device 0's (a box shows all 8 cards of the host; only one is ours)."""
    want = pci_slot_of_hip_device()
    cands = []
    for card in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        slot = None
        try:
            with open(os.path.join(card, 'uevent')) as f:
                for ln in f:
                    if ln.startswith('PCI_SLOT_NAME='):
                        slot = ln.strip().split('=')[1].lower()
        except Exception:
            pass
        hw = sorted(glob.glob(os.path.join(card, 'hwmon', 'hwmon*')))
        if not hw:
            continue
        p = first_readable([os.path.join(hw[0], n) for n in ('power1_average', 'power1_input')])
        if p is None:
            continue
        cands.append({'card': card, 'pci': slot, 'power': p, 'cap': first_readable([os.path.join(hw[0], 'power1_cap')]),
                      'freq': first_readable([os.path.join(hw[0], 'freq1_input')]), 'dpm': first_readable([os.path.join(card, 'pp_dpm_sclk')])})
    for c in cands:
        if want and c['pci'] == want.lower():
            c['matched_by'] = 'pci address of HIP device 0 (%s)' % want
            return c
    if cands:
        cands[0]['matched_by'] = 'first readable card (HIP device 0 is %s, cards seen: %s)' % (want, [c['pci'] for c in cands])
        return cands[0]
    return {}


def read_num(path, scale):
    try:
        with open(path) as f:
            return float(f.read().strip()) * scale
    except Exception:
        return None


def read_dpm(path):
    try:
        with open(path) as f:
            for ln in f:
                if '*' in ln:
                    return float(ln.split(':')[1].replace('Mhz', '').replace('MHz', '').replace('*', '').strip())
    except Exception:
        pass
    return None


class Sampler(threading.Thread):
    def __init__(self, src):
        super().__init__(daemon=True)
        self.src, self.rows, self.stop = src, [], False

    def run(self):
        t0 = time.time()
        while not self.stop:
            row = {'t': round(time.time() - t0, 3)}
            if self.src.get('power'):
                row['power_w'] = read_num(self.src['power'], 1e-6)
            if self.src.get('freq'):
                row['sclk_mhz_hwmon'] = read_num(self.src['freq'], 1e-6)
            if self.src.get('dpm'):
                row['sclk_mhz_dpm'] = read_dpm(self.src['dpm'])
            self.rows.append(row)
            time.sleep(0.05)


def summarise(rows, key):
    v = [r[key] for r in rows if r.get(key) is not None]
    if not v:
        return None
    v.sort()
    return {'n': len(v), 'min': v[0], 'median': v[len(v) // 2], 'max': v[-1], 'mean': sum(v) / len(v)}


src = sysfs_sources()
report = {'sysfs': src, 'B': B, 'half': HALF, 'seconds_per_phase': SECONDS}
if src.get('cap'):
    report['power_cap_w'] = read_num(src['cap'], 1e-6)
if not src:
    try:
        r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=30)
        report['rocm_smi_idle'] = r.stdout[-2000:]
    except Exception as exc:
        report['rocm_smi_idle'] = 'unavailable: %r' % (exc,)

ht = E.HALF_F16 if HALF == 'f16' else E.HALF_BF16
sd = synth.wavenet_state_dict(1234)
eng = E.Engine(max_batch=B, precision=E.BF16, half_type=ht, with_classifier=False)
eng.load_wavenet(sd)
# the same binary on all-zero operands: weight-norm gains, biases and linear layers zeroed (v kept: the fold divides by |v|), so
# every folded weight, the step embedding, the residual stream and the gate are exactly zero
eng0 = E.Engine(max_batch=B, precision=E.BF16, half_type=ht, with_classifier=False)
eng0.load_wavenet({k: (v if k.endswith('weight_v') else v * 0) for k, v in sd.items()})
x = torch.randn(B, 16000, device='cuda') * 0.3
engs = {'random': eng, 'zeros': eng0}
phases = [('idle', None), ('random', x), ('zeros', torch.zeros_like(x))]
if os.environ.get('MIX', '0') == '1':
    LAYER = 'residual_layer.residual_blocks.5.'
    is_l5w = lambda k: k.startswith(LAYER) and ('dilated_conv_layer' in k or 'res_conv' in k) and k.endswith('weight_g')      # noqa: E731
    # layer 5's folded dilated / res weights zero, everything else as trained: random activations into zero weights
    engs['zero_weights'] = E.Engine(max_batch=B, precision=E.BF16, half_type=ht, with_classifier=False)
    engs['zero_weights'].load_wavenet({k: (v * 0 if is_l5w(k) else v) for k, v in sd.items()})
    # only layer 5's weights (and its biases) kept: the stream the timed layer reads is exactly zero, its weights random
    engs['zero_activations'] = E.Engine(max_batch=B, precision=E.BF16, half_type=ht, with_classifier=False)
    engs['zero_activations'].load_wavenet({k: (v if (k.endswith('weight_v') or (k.startswith(LAYER) and 'fc_t' not in k)) else v * 0) for k, v in sd.items()})
    phases += [('zero_weights', x), ('zero_activations', torch.zeros_like(x))]
for name, xin in phases:
    smp = Sampler(src)
    if xin is None:
        smp.start(); time.sleep(2.0); smp.stop = True; smp.join()
        report[name] = {'power_w': summarise(smp.rows, 'power_w'), 'sclk_mhz_hwmon': summarise(smp.rows, 'sclk_mhz_hwmon'),
                        'sclk_mhz_dpm': summarise(smp.rows, 'sclk_mhz_dpm')}
        continue
    en = engs[name]
    eps = en.wavenet_eps(xin, 65)                 # fills the residual stream the timed layer reads
    torch.cuda.synchronize()
    if name == 'zeros':
        assert float(eps.abs().max()) == 0.0
    ms = en.time_layer(5, B, 20)
    iters = max(20, int(SECONDS * 1e3 / ms))
    smp.start()
    t0 = time.time()
    ms = en.time_layer(5, B, iters)
    wall = time.time() - t0
    smp.stop = True; smp.join()
    steady = [r for r in smp.rows if r['t'] > 1.0]          # after the first second under load
    report[name] = {'ms_per_launch': ms, 'launches': iters, 'wall_s': wall, 'tflops': 2.0 * 16000 * (512 * 768 + 256 * 256) * B / (ms * 1e-3) / 1e12,
                    'power_w': summarise(steady, 'power_w'), 'sclk_mhz_hwmon': summarise(steady, 'sclk_mhz_hwmon'),
                    'sclk_mhz_dpm': summarise(steady, 'sclk_mhz_dpm'), 'trace': smp.rows[::4]}
    pw = report[name]['power_w']
    if pw:
        report[name]['joules_per_launch'] = pw['median'] * ms * 1e-3
        report[name]['joules_per_launch_above_idle'] = (pw['median'] - (report['idle']['power_w'] or {}).get('median', 0.0)) * ms * 1e-3
    print(name, json.dumps({k: v for k, v in report[name].items() if k != 'trace'}), flush=True)

# in-kernel clock of the stamped build right after the load phase (stderr line "[dmad stamps] ... in-kernel clock")
eng.wavenet_eps(x, 65); torch.cuda.synchronize()
eng.time_layer(5, B, max(20, int(2.0 * 1e3 / report['random']['ms_per_launch'])))       # 2 s of load first
os.environ['DMAD_LAYER_STAMPS'] = '1'
sys.stderr.flush()
ms_st = eng.time_layer(5, B, 10)
report['stamped_build_ms_per_launch'] = ms_st
eng0.time_layer(5, B, max(20, int(2.0 * 1e3 / report['zeros']['ms_per_launch'])))
report['stamped_build_ms_per_launch_zeros'] = eng0.time_layer(5, B, 10)      # second "[dmad stamps]" line: the clock on zero operands
with open(os.path.join(OUT, 'power_trace.json'), 'w') as fh:
    json.dump(report, fh, indent=1)
print(json.dumps({k: v for k, v in report.items() if k not in engs}), flush=True)
for en in engs.values():
    en.close()
