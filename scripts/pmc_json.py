#!/usr/bin/env python3
"""Turn the three PMC passes of scripts/gpu_session.sh (FETCH_SIZE | WRITE_SIZE | SQ counters; bench.py --headline-only, so
every launch of the fitness kernels has the headline's shape) into profiles/pmc_latest.json.  The record carries the hash
of the kernel sources and the trees per launch: bench.py quotes `traffic` only when both match what it runs.

    python scripts/pmc_json.py pmc1.md pmc2.md pmc3.md bench_line.log > profiles/pmc_latest.json
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def table(path):
    out = {}
    for line in open(path):
        m = re.match(r"\| `(.+?)` \| (\w+) \| (\d+) \| ([\d.e+-]+) \| ([\d.e+-]+) \|", line)
        if m:
            out.setdefault(m.group(1), {})[m.group(2)] = (int(m.group(3)), float(m.group(5)))
    return out


def main():
    import bench

    t = {}
    for p in sys.argv[1:4]:
        for k, v in table(p).items():
            t.setdefault(k, {}).update(v)
    line = None
    for raw in open(sys.argv[4]):
        raw = raw.strip()
        if raw.startswith("{") and '"metric"' in raw:
            line = json.loads(raw)
    interp = next((k for k in t if "sr_tc_kernel<8, false, 2" in k), None)   # (<K, STATS, FAST[, WIDE]>: the short-division 8-row build)
    # the program compiler of the headline: the packed kernel (round 4), else the one-tree kernel -- never the general compiler's launch
    comp = next((k for k in t if "tc_compile_packed_kernel" in k), None) or next((k for k in t if "tc_compile_kernel" in k), None)
    out = {
        "what": "rocprofv3 --kernel-trace --pmc, one counter set per pass (FETCH_SIZE | WRITE_SIZE | SQ_*), command: python bench.py "
                "--steps 4 --warmup 1 --headline-only; per-dispatch averages (scripts/rocpd_summary.py)",
        "source_sha": bench.source_sha(),
        "pop_per_launch": line["config"]["pop_per_gpu"] if line else None,
        "files": [os.path.basename(p) for p in sys.argv[1:4]],
        "kernels": {},
    }
    # rocprofv3 stores one row per launch AND counter instance for the SQ counters (32 rows per launch on this device: the
    # one-launch xcc_probe_kernel shows it), each holding that instance's share; the derived TCC sizes are per launch already
    probe = next((k for k in t if "xcc_probe" in k), None)
    inst = max(v[0] for v in t[probe].values()) if probe else 32
    out["sq_rows_per_launch"] = inst
    for name, key in ((interp, "sr_tc_kernel"), (comp, "tc_compile_kernel")):
        if not name:
            continue
        c = {k: (v[1] * inst if k.startswith("SQ_") else v[1]) for k, v in t[name].items()}
        out["kernels"][key] = {"rocprof_name": name, "launches": max(v[0] for v in t[name].values()) // inst, "per_launch": c}
    if interp and "FETCH_SIZE" in t[interp] and "WRITE_SIZE" in t[interp]:
        f, w = t[interp]["FETCH_SIZE"][1] * 1024, t[interp]["WRITE_SIZE"][1] * 1024
        pop = out["pop_per_launch"] or 0
        out["sr_tc_kernel_fetch_bytes_raw"] = f
        out["sr_tc_kernel_write_bytes_raw"] = w
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KB) reports half the bytes of wide coalesced reads (16 B per lane).  The interpreter's
        # HBM reads are of exactly that kind (the global_load_dwordx4 warm-up loads pull the records into L2, the scalar loads then
        # hit L2), and the known byte count calibrates it: pop x 256-byte records.
        out["sr_tc_kernel_hbm_bytes_per_launch"] = 2 * f + w
        out["correction"] = ("counter values are KB (x 1024); FETCH_SIZE doubled (gfx950 tallies 128-byte requests of 16 B/lane coalesced reads at 64 B; "
                             f"calibration on this kernel: records = pop x 256 B = {pop * 256} B, raw FETCH_SIZE = {f:.0f} B); WRITE_SIZE as reported")
    if comp and "FETCH_SIZE" in t.get(comp, {}) and "WRITE_SIZE" in t[comp] and "sr_tc_kernel_hbm_bytes_per_launch" in out:
        # the whole fitness call: + the program compiler, which reads the forest (2- and 4-byte loads per lane) and writes the records the
        # interpreter then reads.  Its FETCH_SIZE is doubled too: calibrated in round 4 with reads of a known byte count in exactly these
        # widths (scripts/pmc_calibrate.py, profiles/r04P_06_calib.md: 2, 4 and 16 bytes per lane over 2 GiB all report 1.0486e6 KB,
        # exactly one half)
        cf, cw = t[comp]["FETCH_SIZE"][1] * 1024, t[comp]["WRITE_SIZE"][1] * 1024
        out["tc_compile_kernel_fetch_bytes_raw"] = cf
        out["tc_compile_kernel_write_bytes_raw"] = cw
        out["tc_compile_kernel_hbm_bytes_per_launch"] = 2 * cf + cw
        out["call_hbm_bytes"] = out["sr_tc_kernel_hbm_bytes_per_launch"] + 2 * cf + cw
        out["correction"] += ("; the compiler's FETCH_SIZE doubled as well (calibrated: 2-, 4- and 16-byte-per-lane reads of 2 GiB each report exactly "
                              "half, profiles/r04P_06_calib.md)")
    if interp and "SQ_WAVE_CYCLES" in t[interp]:
        c = {k: v[1] * inst for k, v in t[interp].items()}
        wc = c["SQ_WAVE_CYCLES"]
        out["sq"] = {k.lower() + "_over_wave_cycles": c[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in c}
        if "SQ_INSTS_VALU" in c and "SQ_INSTS_SALU" in c:
            out["sq"]["insts_valu_over_insts_salu"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_SALU"]
            out["sq"]["insts_valu_per_launch"] = c["SQ_INSTS_VALU"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
