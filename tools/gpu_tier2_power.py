#!/usr/bin/env python3
"""Evidence for the split-f16 tier: board power and sclk while WaveNet evaluations of one path run back to
back for SECONDS (hwmon of the card whose PCI address is HIP device 0's, 50 ms samples — the sampler of tools/gpu_power_trace.py).
DMAD_LIB selects the library build (the ablation builds of the tier's kernel are compared this way); PATH_ID: 2 = split-f16 tier,
1 = exact fp32, 0 = 16-bit.      B=19 SECONDS=5 PATH_ID=2 python tools/gpu_tier2_power.py   -> one JSON line"""
import glob, json, os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'diffusion-model-for-audio-defense_amd')]
from dmad_hip import _lib
if os.environ.get('DMAD_LIB'):
    _lib.LIB_PATH = os.environ['DMAD_LIB']
from dmad_hip import engine as E, synth  # noqa: E402

B, SECONDS, PATH_ID = int(os.environ.get('B', 19)), float(os.environ.get('SECONDS', 5)), int(os.environ.get('PATH_ID', 2))


def hwmon_of_device0():
    p = torch.cuda.get_device_properties(0)
    want = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    for card in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        try:
            slot = [ln.strip().split('=')[1].lower() for ln in open(os.path.join(card, 'uevent')) if ln.startswith('PCI_SLOT_NAME=')][0]
        except Exception:
            continue
        hw = sorted(glob.glob(os.path.join(card, 'hwmon', 'hwmon*')))
        if slot == want and hw:
            return hw[0], want
    return None, want


def num(path, scale):
    try:
        return float(open(path).read().strip()) * scale
    except Exception:
        return None


hw, pci = hwmon_of_device0()
rows, stop = [], [False]


def sample():
    while not stop[0]:
        if hw:
            rows.append((num(os.path.join(hw, 'power1_average'), 1e-6) or num(os.path.join(hw, 'power1_input'), 1e-6), num(os.path.join(hw, 'freq1_input'), 1e-6)))
        time.sleep(0.05)


eng = E.Engine(max_batch=B, precision=E.EXACT, recheck_batch=B, with_classifier=False)
eng.load_wavenet(synth.wavenet_state_dict(1234))
x = torch.randn(B, 16000, device='cuda') * 0.3
eng.wavenet_eps_path(x, 65, PATH_ID); torch.cuda.synchronize()
th = threading.Thread(target=sample, daemon=True); th.start()
time.sleep(0.3)
n_idle = len(rows)
t0 = time.time(); it = 0
while time.time() - t0 < SECONDS:
    eng.wavenet_eps_path(x, 65, PATH_ID); it += 1
    if it % 4 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
dt = time.time() - t0
stop[0] = True; th.join()
load = rows[n_idle + len(rows[n_idle:]) // 3:]                    # the last two thirds of the loaded phase
med = lambda v: sorted(v)[len(v) // 2] if v else None
print(json.dumps({'lib': os.environ.get('DMAD_LIB', 'in-tree'), 'path': PATH_ID, 'B': B, 'pci': pci, 'hwmon': hw,
                  'cap_w': num(os.path.join(hw, 'power1_cap'), 1e-6) if hw else None,
                  'ms_per_evaluation': round(dt / it * 1e3, 2), 'clips_per_s': round(B * it / dt, 1),
                  'power_w_median': med([r[0] for r in load if r[0] is not None]), 'sclk_mhz_median': med([r[1] for r in load if r[1] is not None]),
                  'power_w_idle': med([r[0] for r in rows[:n_idle] if r[0] is not None])}))
