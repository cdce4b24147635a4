"""Board power / sclk sampler shared by the development tools (hwmon of the card whose PCI address is HIP device 0's).

    src = sysfs_sources(); smp = Sampler(src); smp.start(); ...load...; smp.stop = True; smp.join(); summarise(smp.rows, 'power_w')
"""
import glob
import os
import threading
import time

import torch


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
    try:
        p = torch.cuda.get_device_properties(0)
        return '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def sysfs_sources():
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
                      'freq': first_readable([os.path.join(hw[0], 'freq1_input')])})
    for c in cands:
        if want and c['pci'] == want.lower():
            c['matched_by'] = 'pci address of HIP device 0 (%s)' % want
            return c
    if cands:
        cands[0]['matched_by'] = 'first readable card (HIP device 0 is %s)' % want
        return cands[0]
    return {}


def read_num(path, scale):
    try:
        with open(path) as f:
            return float(f.read().strip()) * scale
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self, src, period=0.05):
        super().__init__(daemon=True)
        self.src, self.rows, self.stop, self.period = src, [], False, period

    def run(self):
        t0 = time.time()
        while not self.stop:
            row = {'t': round(time.time() - t0, 3)}
            if self.src.get('power'):
                row['power_w'] = read_num(self.src['power'], 1e-6)
            if self.src.get('freq'):
                row['sclk_mhz'] = read_num(self.src['freq'], 1e-6)
            self.rows.append(row)
            time.sleep(self.period)


def summarise(rows, key):
    v = sorted(r[key] for r in rows if r.get(key) is not None)
    if not v:
        return None
    return {'n': len(v), 'min': v[0], 'median': v[len(v) // 2], 'max': v[-1], 'mean': sum(v) / len(v)}


def measure(fn, src, settle=1.0):
    """run fn() under the sampler -> (fn's result, median power after `settle` seconds, median sclk)."""
    smp = Sampler(src)
    smp.start()
    out = fn()
    smp.stop = True
    smp.join()
    rows = [r for r in smp.rows if r['t'] > settle] or smp.rows
    pw, fq = summarise(rows, 'power_w'), summarise(rows, 'sclk_mhz')
    return out, (pw or {}).get('median'), (fq or {}).get('median')
