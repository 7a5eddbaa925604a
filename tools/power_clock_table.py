"""The held shader clock as a design variable: one process, one box, one table

    {variant, W (socket power), GHz (shader clock inside the load), TFLOP/s, modelled HBM GB per fit}

for (i) the register-only f64-MFMA probe, (ii) the headline fit! + predict step (two leaf lanes), (iii) the same with every Gram
tile through memory first (DSMGP_OPT_FUSED_GRAM 0: fewer fp64 exp inside the update launches, more HBM), (iv) one lane,
(v) the update kernel on uniform batches with its operands streamed from HBM and with all operands L2-resident (diagnostic build,
tools/steady_state_tile.py's modes 0 / 1: the same instruction stream at full and at (nearly) no HBM traffic).

Power and clock come from the amdgpu hwmon files of the card under load (power1_average / power1_input in microwatts, freq1_input
in Hz), read by a side thread at ~50 Hz; where the box does not expose them to an ordinary user the thread falls back to
`amd-smi metric` / `rocm-smi` in a loop (a few Hz).  The shader clock INSIDE the kernels is the library's own one-wave sampler
(dsmgp_clock_sample_*).  Numbers only: nothing here is used by the product.

    python tools/power_clock_table.py > gpurun_out/power_clock.json
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Sampler:
    """Side thread: (t, {card: watts}, {card: sclk MHz}) at ~50 Hz from sysfs, or a few Hz from amd-smi / rocm-smi."""

    def __init__(self):
        self.power_files, self.freq_files = {}, {}
        for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            card = hw.split("/")[4]
            for name in ("power1_average", "power1_input"):
                p = os.path.join(hw, name)
                if os.path.exists(p) and card not in self.power_files:
                    try:
                        int(open(p).read())
                        self.power_files[card] = p
                    except Exception:
                        pass
            p = os.path.join(hw, "freq1_input")
            if os.path.exists(p):
                try:
                    int(open(p).read())
                    self.freq_files[card] = p
                except Exception:
                    pass
        self.caps = {}
        for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            for name in ("power1_cap", "power1_cap_max", "power1_cap_default"):
                try:
                    self.caps.setdefault(hw.split("/")[4], {})[name] = int(open(os.path.join(hw, name)).read()) / 1e6
                except Exception:
                    pass
        self.mode = "sysfs" if self.power_files else None
        if self.mode is None:
            for cmd in (["amd-smi", "metric", "--power", "--clock", "--json"], ["rocm-smi", "--showpower", "--showclocks", "--json"]):
                try:
                    out = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
                    if out.returncode == 0 and out.stdout.strip().startswith(("{", "[")):
                        self.mode, self.cmd = "smi", cmd
                        break
                except Exception:
                    pass
        self.samples = []
        self._stop = False
        self.th = None

    def _read_smi(self):
        out = subprocess.run(self.cmd, capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        pw, fq = {}, {}

        def walk(o, card):
            if isinstance(o, dict):
                for k, v in o.items():
                    kl = str(k).lower()
                    if kl in ("gpu",) and isinstance(v, (int, str)):
                        card = f"gpu{v}"
                    if isinstance(v, (dict, list)):
                        walk(v, card if not kl.startswith("card") else kl)
                    else:
                        try:
                            val = float(str(v).split()[0])
                        except Exception:
                            continue
                        if "power" in kl and "cap" not in kl and "limit" not in kl and val > 0:
                            pw.setdefault(card or "gpu", val)
                        if ("sclk" in kl or kl in ("clk", "gfx_0")) and val > 0:
                            fq.setdefault(card or "gpu", val)
            elif isinstance(o, list):
                for i, v in enumerate(o):
                    walk(v, card or f"gpu{i}")
        walk(j, None)
        return pw, fq

    def _run(self):
        while not self._stop:
            t = time.perf_counter()
            try:
                if self.mode == "sysfs":
                    pw = {c: int(open(p).read()) / 1e6 for c, p in self.power_files.items()}
                    fq = {c: int(open(p).read()) / 1e6 for c, p in self.freq_files.items()}
                else:
                    pw, fq = self._read_smi()
                self.samples.append((t, pw, fq))
            except Exception:
                pass
            if self.mode == "sysfs":
                time.sleep(0.02)

    def start(self):
        if self.mode is None:
            return
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def stop(self):
        self._stop = True
        if self.th is not None:
            self.th.join(timeout=30)

    def window(self, t0, t1, card=None):
        """mean / max watts and mean sclk of the samples inside [t0, t1] -- of `card`, or of the card that draws most there."""
        sel = [s for s in self.samples if t0 <= s[0] <= t1]
        if not sel:
            return None
        cards = sorted({c for s in sel for c in s[1]})
        if not cards:
            return None
        mean = {c: float(np.mean([s[1][c] for s in sel if c in s[1]])) for c in cards}
        card = card or max(mean, key=mean.get)
        w = [s[1][card] for s in sel if card in s[1]]
        f = [s[2][card] for s in sel if card in s[2]]
        return {"card": card, "samples": len(w), "watts_mean": float(np.mean(w)), "watts_max": float(np.max(w)),
                "sclk_mhz_mean": float(np.mean(f)) if f else None}


def main():
    import torch  # noqa: F401  (device memory / streams of the package)
    import bench
    import deepstructuredmixtures_amd as dsm
    from deepstructuredmixtures_amd import hipabi

    smp = Sampler()
    smp.start()
    out = {"sampler": smp.mode, "power_files": smp.power_files, "power_caps_watts": smp.caps, "variants": []}
    time.sleep(1.5)
    idle = smp.window(time.perf_counter() - 1.5, time.perf_counter())
    out["idle"] = idle

    model, X, y, Xt, ptr, idx = bench.build_model("dsmgp_n100k_d8", 0, 1, 0)
    ctx = model.ctx
    card = [None]
    try:    # the card this process computes on, by PCI address (the box's sysfs shows every card of the host, other tenants' too)
        pr = torch.cuda.get_device_properties(0)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        for c_ in smp.power_files:
            if os.path.basename(os.path.realpath(f"/sys/class/drm/{c_}/device")).startswith(want):
                card[0] = c_
        out["pci"] = want
    except Exception as e:      # noqa: BLE001
        out["pci_error"] = str(e)
    out["card"] = card[0]
    if card[0]:
        out["idle"] = smp.window(0, time.perf_counter(), card[0])

    def loaded(name, fn, seconds, extra=None):
        """Run fn() repeatedly for `seconds`, the library's clock sampler across the middle of it; -> one table row."""
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps, res, ghz = 0, None, []
        while time.perf_counter() - t0 < seconds:
            res = fn(sample=True)
            if isinstance(res, dict) and res.get("ghz"):
                ghz.append(res["ghz"])
            reps += 1
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        w = smp.window(t0 + 0.3 * (t1 - t0), t1, card[0])     # the first third: ramp
        if w and card[0] is None:
            card[0] = w["card"]
        row = {"variant": name, "seconds": t1 - t0, "repetitions": reps, "power": w,
               "clock_ghz_in_kernels": float(np.median(ghz)) if ghz else None}
        if isinstance(res, dict):
            row.update({k: v for k, v in res.items() if k != "ghz"})
        if extra:
            row.update(extra)
        out["variants"].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)

    # (i) register-only probe
    def probe(sample=False):
        p = ctx.probe_f64_mfma_detail(8)
        return {"tflops": p["tflops"], "ghz": p.get("clock_ghz")}
    loaded("f64 MFMA probe (register-only, 8 waves per SIMD)", probe, 6.0, {"hbm_gb_per_fit_modelled": 0.0})

    # (ii)-(iv) the headline step
    def make_step():
        dsm.resident_test(model, Xt)
        ctx.set_profile(1)
        st = {"t": 0.4}

        def step(sample=False):
            if sample:
                ctx.clock_sample_start(min(4000.0, 800.0 * st["t"]))
            ts = time.perf_counter()
            dsm.fit(model)
            dsm.update(model)
            dsm.predict(model, Xt)
            st["t"] = time.perf_counter() - ts
            tm = ctx.timings()
            un = tm.get("chol_update_union", 0.0) or tm.get("chol_update", 0.0)
            fl, nl = ctx.work()
            r = {"step_s": st["t"], "update_union_s": un, "tflops": fl / un / 1e12 if un > 0 else None, "update_launches": nl,
                 "lanes": ctx.lanes()}
            if sample:
                r["ghz"] = ctx.clock_sample_read()[0]
            return r
        return step

    nobs = np.array([lf.nobs for lf in model.leaves], dtype=np.float64)
    npad = np.ceil(nobs / 128) * 128
    # operand bytes of the update launches when every leaf's B panel is read once per step and every tile's A panel once
    # (796 GB) -- the counters read 956 GB (profiles/r05_update_kernel_traffic.json); unfused Gram adds one write and
    # one read of every lower tile: 8 n^2 / 2 each
    gram_extra = float(np.sum(npad ** 2) / 2 * 8 * 2 / 1e9)
    loaded("headline step, two lanes (as benched)", make_step(), 8.0, {"hbm_gb_per_fit_modelled": 956.0})
    model.set_option(hipabi.OPT_FUSED_GRAM, 0)
    loaded("headline step, Gram tiles through memory (OPT_FUSED_GRAM 0)", make_step(), 8.0,
           {"hbm_gb_per_fit_modelled": 956.0 + gram_extra})
    model.set_option(hipabi.OPT_FUSED_GRAM, 1)
    model.set_option(hipabi.OPT_LANES, 1)
    loaded("headline step, one lane", make_step(), 8.0, {"hbm_gb_per_fit_modelled": 836.0})
    model.set_option(hipabi.OPT_LANES, 0)
    ctx.close()

    # (v) the update kernel alone, operands from HBM against operands L2-resident (diagnostic build)
    try:
        dctx = hipabi.Context(0, diag=True)
        for mode, label, gb in ((0, "update kernel, uniform batch 4096 tiles K=4096, operands streamed (own A panel per tile, B shared by 16)", None),
                                (1, "update kernel, same batch, ALL operands shared (L2-resident: ~no HBM traffic)", 0.0)):
            def tile(sample=False, mode=mode):
                if sample:
                    dctx.clock_sample_start(600.0)
                tf = dctx.bench_tile(4096, 4096, mode, 16, 120)     # ~1 s of back-to-back launches per call
                r = {"tflops": tf}
                if sample:
                    r["ghz"] = dctx.clock_sample_read()[0]
                return r
            a_bytes = 4096 * 128 * 4096 * 8 * (1 + 1 / 16) / 1e9
            loaded(label, tile, 6.0, {"hbm_gb_per_launch_modelled": a_bytes if mode == 0 else 0.0})
        dctx.close()
    except Exception as e:      # noqa: BLE001
        out["diag_error"] = str(e)
    smp.stop()
    out["n_samples"] = len(smp.samples)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
