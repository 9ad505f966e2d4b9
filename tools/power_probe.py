#!/usr/bin/env python
"""Socket power and shader clock of the GPU while ONE workload loops -- the measurement behind (or against) "these launches sit at
the power envelope" (DESIGN 4e).  A sampler thread reads the amdgpu hwmon files of the card every few ms while the main thread keeps
the workload running for --seconds; power comes from power1_average (falling back to power1_input), the clock from freq1_input,
the cap from power1_cap.  Nothing here needs root; when the files are missing the script falls back to `rocm-smi --showpower
--showclocks --json` polled from a child (coarser) and says so.

Usage:  python tools/power_probe.py [--seconds 3] [--only name,name]
Workloads: idle, gemm_f16 (torch / hipBLASLt fp16 8192^3 GEMM: a dense 16-bit MFMA stream, sustained), hbm_copy (1 GiB device copy),
           apply (the bench headline launch), f16x3_64 / f16x3_256 / f16x3_51 / f16x3_32 (one F16X3 layer each), x6_64, fp32_64,
           mfma_chain (tools/micro/mfma_chain as a child process: v_mfma_f32_32x32x16_f16 back to back at 2.0-2.2 PFLOP/s, but in 1.5 ms
           bursts between process starts -- its watts are a duty-cycle average, not the draw of a sustained stream: see gemm_f16)
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sstem-restoration_amd"))


def find_hwmon(pci=None):
    """hwmon directory of the card HIP runs on.  A box of the pool shows every card of its host under /sys while one is visible to HIP:
    the directory is looked up through the PCI address torch reports for device 0 (the first probe of this script read card0, another
    tenant's GPU: 286 W "idle"); without one, every amdgpu hwmon is returned and the caller picks the card whose power moves."""
    cands = []
    if pci:
        cands = sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % pci))
    if not cands:
        cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    out = []
    for d in cands:
        try:
            if open(os.path.join(d, "name")).read().strip() != "amdgpu":
                continue
        except OSError:
            continue
        have = {f: os.path.exists(os.path.join(d, f)) for f in ("power1_average", "power1_input", "freq1_input", "power1_cap")}
        if not (have["power1_average"] or have["power1_input"]):
            continue
        try:
            open(os.path.join(d, "power1_input" if have["power1_input"] else "power1_average")).read()
        except OSError:
            continue
        out.append((d, have))
    return out


class Sampler(threading.Thread):
    def __init__(self, hw, period=0.004):
        super().__init__(daemon=True)
        self.d, self.have = hw
        self.period = period
        self.stop = False
        self.rows = []

    def _read(self, f):
        try:
            return float(open(os.path.join(self.d, f)).read())
        except (OSError, ValueError):
            return float("nan")

    def run(self):
        pf = "power1_input" if self.have["power1_input"] else "power1_average"
        while not self.stop:
            self.rows.append((time.perf_counter(), self._read(pf) / 1e6, self._read("freq1_input") / 1e6 if self.have["freq1_input"] else float("nan")))
            time.sleep(self.period)


def summarise(rows, t0, t1):
    # the first third of the window is ramp (clocks, the power average's own filter): report the rest
    lo = t0 + (t1 - t0) / 3.0
    w = [r[1] for r in rows if lo <= r[0] <= t1 and r[1] == r[1]]
    f = [r[2] for r in rows if lo <= r[0] <= t1 and r[2] == r[2]]
    if not w:
        return None
    return {"watts_mean": round(sum(w) / len(w), 1), "watts_max": round(max(w), 1), "mhz_mean": round(sum(f) / len(f), 0) if f else None,
            "samples": len(w)}


def smi_sample():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
        return json.loads(r.stdout)
    except Exception as exc:       # noqa: BLE001
        return {"error": str(exc)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    import hipnn.functional as HF
    from libs.sepconv.fused import coef_to_blocked, interp_apply_gray_blocked
    dev = torch.device("cuda")
    pr = torch.cuda.get_device_properties(0)
    pci = None
    if hasattr(pr, "pci_bus_id"):
        pci = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    hws = find_hwmon(pci)
    if len(hws) > 1:
        # no PCI address to go by: load the GPU for a second and take the card whose power reading rises most
        def read(h):
            try:
                return float(open(os.path.join(h[0], "power1_input" if h[1]["power1_input"] else "power1_average")).read())
            except (OSError, ValueError):
                return 0.0
        torch.cuda.synchronize(); time.sleep(0.5)
        before = [read(h) for h in hws]
        x = torch.randn(8192, 8192, device=dev)
        t_ = time.perf_counter()
        while time.perf_counter() - t_ < 1.5:
            (x @ x).sum().item()
        after = [read(h) for h in hws]
        hws = [hws[max(range(len(hws)), key=lambda i: after[i] - before[i])]]
        del x
    hw = hws[0] if hws else None
    res = {"hwmon": hw[0] if hw else None, "pci": pci, "power_file": ("power1_input" if hw[1]["power1_input"] else "power1_average") if hw else None}
    if hw:
        try:
            res["power_cap_w"] = float(open(os.path.join(hw[0], "power1_cap")).read()) / 1e6
        except (OSError, ValueError):
            pass
    print("hwmon:", res, flush=True)

    def conv_layer(N, C, S, algo, Cout=None):
        x = torch.randn(N, C, S, S, device=dev)
        w = torch.randn(Cout or C, C, 3, 3, device=dev) * 0.05
        b = torch.randn(Cout or C, device=dev)
        owner = torch.nn.Module()            # keeps the packed weights between calls, as an inference layer does

        def fn():
            with HF.algorithm(algo), torch.no_grad():
                HF.conv2d_fused(x, w, b, None, None, HF.ACT_RELU, 0.0, owner=owner)
        flop = 2.0 * N * (Cout or C) * S * S * C * 9
        return fn, flop

    def apply_wl():
        g = torch.Generator(device=dev); g.manual_seed(555)
        B, S = 8, 1024
        i1 = torch.rand(B, 1, S, S, device=dev, generator=g); i2 = torch.rand(B, 1, S, S, device=dev, generator=g)
        ks = [coef_to_blocked(torch.softmax(torch.randn(B, 51, S, S, device=dev, generator=g), 1)) for _ in range(4)]

        def fn():
            with torch.no_grad():
                interp_apply_gray_blocked(i1, i2, *ks)
        return fn, None

    def gemm_f16():
        a_ = torch.randn(8192, 8192, device=dev, dtype=torch.float16); b_ = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
        return (lambda: torch.matmul(a_, b_)), 2.0 * 8192 ** 3

    def hbm_copy():
        a_ = torch.empty(1 << 28, device=dev); b_ = torch.randn(1 << 28, device=dev)       # 1 GiB read + 1 GiB written per call
        return (lambda: a_.copy_(b_)), None

    def sepconv_wl(kind):
        import libs.sepconv._ext.cunnex as cunnex
        g = torch.Generator(device=dev); g.manual_seed(556)
        B, S = 8, 1024
        rgb = kind.endswith("rgb")
        inp = torch.rand(B, 3 if rgb else 1, S + 50, S + 50, device=dev, generator=g).expand(B, 3, S + 50, S + 50).contiguous()
        ver = torch.softmax(torch.randn(B, 51, S, S, device=dev, generator=g), 1); hor = torch.softmax(torch.randn(B, 51, S, S, device=dev, generator=g), 1)
        out = torch.empty(B, 3, S, S, device=dev); gout = torch.randn(B, 3, S, S, device=dev, generator=g)
        gv, gh = torch.empty_like(ver), torch.empty_like(hor)
        if kind.startswith("fwd"):
            return (lambda: cunnex.SeparableConvolution_cuda_forward(inp, ver, hor, out)), None
        return (lambda: cunnex.SeparableConvolution_cuda_backward(gout, inp, ver, hor, None, gv, gh)), None

    workloads = {
        "idle": lambda: (None, None),
        "sepconv_fwd_gray": lambda: sepconv_wl("fwd_gray"), "sepconv_fwd_rgb": lambda: sepconv_wl("fwd_rgb"),
        "sepconv_bwd_gray": lambda: sepconv_wl("bwd_gray"), "sepconv_bwd_rgb": lambda: sepconv_wl("bwd_rgb"),
        "gemm_f16": gemm_f16,
        "hbm_copy": hbm_copy,
        "apply": apply_wl,
        "f16x3_64": lambda: conv_layer(8, 64, 512, HF.ALGO_MFMA_F16X3),
        "f16x3_256": lambda: conv_layer(8, 256, 128, HF.ALGO_MFMA_F16X3),
        "f16x3_51": lambda: conv_layer(8, 51, 1024, HF.ALGO_MFMA_F16X3),
        "f16x3_32": lambda: conv_layer(8, 32, 1024, HF.ALGO_MFMA_F16X3),
        "x6_64": lambda: conv_layer(8, 64, 512, HF.ALGO_MFMA_BF16X6),
        "fp32_64": lambda: conv_layer(8, 64, 512, HF.ALGO_MFMA),
        "mfma_chain": None,
    }
    names = a.only.split(",") if a.only else list(workloads)
    rows = {}
    for name in names:
        sampler = Sampler(hw) if hw else None
        if sampler:
            sampler.start()
        time.sleep(0.3)
        t0 = time.perf_counter()
        extra = {}
        if name == "mfma_chain":
            exe = os.path.join(REPO, "tools", "micro", "mfma_chain")
            if not os.path.exists(exe):
                rows[name] = {"error": "tools/micro/mfma_chain not built"}
                continue
            n = 0
            outp = ""
            while time.perf_counter() - t0 < a.seconds:
                outp = subprocess.run([exe], capture_output=True, text=True).stdout
                n += 1
            extra["child_runs"] = n
            extra["last_line"] = outp.strip().splitlines()[-1] if outp.strip() else None
        elif name == "idle":
            torch.cuda.synchronize()
            time.sleep(a.seconds)
        else:
            fn, flop = workloads[name]()
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            while time.perf_counter() - t0 < a.seconds:
                for _ in range(20):
                    fn()
                n += 20
                torch.cuda.synchronize()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            extra["ms_per_launch"] = round(ms, 4)
            if flop:
                extra["tflops_fp32_equiv"] = round(flop / ms / 1e9, 1)
        t1 = time.perf_counter()
        smi = None if hw else smi_sample()
        if sampler:
            sampler.stop = True
            sampler.join()
            s = summarise(sampler.rows, t0, t1) or {}
        else:
            s = {"rocm_smi": smi}
        rows[name] = dict(s, **extra)
        print("%-12s %s" % (name, json.dumps(rows[name])), flush=True)
        torch.cuda.empty_cache()
    res["workloads"] = rows
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
