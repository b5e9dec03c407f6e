"""Shader clock and board power of one GPU, sampled from sysfs by a side thread while a timed leg runs.

The matrix kernels of this package are bound by board power: the clock the chip holds under them differs from box to
box by a few per cent, and with it every throughput figure.  A measurement that carries the mean clock and power it
was taken at can be compared with one from another box.  Nothing here touches the HIP runtime: the files are the
amdgpu driver's (`pp_dpm_sclk`: the active level is marked `*`; hwmon `freq1_input` in Hz, `power1_average` /
`power1_input` in microwatts).  A box without readable files yields None fields, never an error.
"""
from __future__ import annotations

import glob
import os
import re
import threading
import time
from typing import Dict, List, Optional


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


def find_card(pci_bus_id: Optional[str] = None) -> Optional[str]:
    """sysfs device directory of the GPU with this PCI address ('0000:c1:00.0'), or of the only amdgpu card."""
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if os.path.exists(os.path.join(dev, "pp_dpm_sclk")) or glob.glob(os.path.join(dev, "hwmon/hwmon*/power1_*")):
            cards.append(dev)
    if pci_bus_id:
        want = pci_bus_id.lower()
        for dev in cards:
            if os.path.basename(os.path.realpath(dev)).lower() == want:
                return dev
    return cards[0] if len(cards) == 1 else None


def device_pci_bus_id(index: int) -> Optional[str]:
    """'dddd:bb:dd.f' of a torch device, from its properties (None when this torch build does not expose them)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        return "{:04x}:{:02x}:{:02x}.0".format(p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


class Sampler:
    """with Sampler(card) as s: ...timed region...; s.summary() -> {"clock_ghz_mean", "power_w_mean", "samples", ...}."""

    def __init__(self, card: Optional[str], period_s: float = 0.05):
        self.card, self.period = card, period_s
        self.clock_mhz: List[float] = []
        self.power_w: List[float] = []
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self._hwmon = sorted(glob.glob(os.path.join(card, "hwmon/hwmon*"))) if card else []

    def _clock(self) -> Optional[float]:
        for h in self._hwmon:
            t = _read(os.path.join(h, "freq1_input"))
            if t and t.strip().isdigit() and int(t) > 0:
                return int(t) / 1e6
        t = _read(os.path.join(self.card, "pp_dpm_sclk"))
        if t:
            m = re.search(r"(\d+)\s*mhz\s*\*", t, re.I)
            if m:
                return float(m.group(1))
        return None

    def _power(self) -> Optional[float]:
        for h in self._hwmon:
            for name in ("power1_average", "power1_input"):
                t = _read(os.path.join(h, name))
                if t and t.strip().isdigit() and int(t) > 0:
                    return int(t) / 1e6
        return None

    def _run(self):
        while not self._stop.is_set():
            c, p = self._clock(), self._power()
            if c is not None:
                self.clock_mhz.append(c)
            if p is not None:
                self.power_w.append(p)
            self._stop.wait(self.period)

    def __enter__(self):
        if self.card:
            self._thread = threading.Thread(target=self._run, name="gpustat", daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        return False

    def summary(self) -> Dict[str, Optional[float]]:
        def mean(v):
            return round(sum(v) / len(v), 1) if v else None
        c = mean(self.clock_mhz)
        return {"clock_ghz_mean": round(c / 1e3, 3) if c else None, "clock_ghz_min": round(min(self.clock_mhz) / 1e3, 3) if self.clock_mhz else None,
                "power_w_mean": mean(self.power_w), "power_w_max": round(max(self.power_w), 1) if self.power_w else None,
                "samples": max(len(self.clock_mhz), len(self.power_w)),
                "source": (self.card + " (sysfs, %d ms period)" % int(1e3 * self.period)) if self.card else None}


def sampler_for(index: int, period_s: float = 0.05) -> Sampler:
    return Sampler(find_card(device_pci_bus_id(index)), period_s)


if __name__ == "__main__":   # what this box exposes: python -m nerf_amd.gpustat
    card = find_card(None)
    print("card:", card)
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        print(dev, "->", os.path.realpath(dev), "sclk:", (_read(os.path.join(dev, "pp_dpm_sclk")) or "").replace("\n", " | "))
        for h in glob.glob(os.path.join(dev, "hwmon/hwmon*")):
            for f in sorted(os.listdir(h)):
                if f.startswith(("freq", "power")):
                    print("   ", os.path.join(h, f), (_read(os.path.join(h, f)) or "").strip())
    with Sampler(card) as s:
        time.sleep(0.5)
    print(s.summary())
