#!/usr/bin/env python
"""Reduces the rocprofv3 --pmc passes (rocpd sqlite) to the per-kernel table kept under profiles/ and to the
pmc_traffic.json that bench.py reads for `roofline.traffic`.

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced read, so it is doubled.  The factor is
re-calibrated here on a kernel with a known byte count in the same access pattern (sinkhorn_sweep reads S once:
B*N*N*4 bytes with 16-B/lane loads; it writes 2 partial planes) - see the `calibration` entry of the JSON.
"""
import json
import os
import sqlite3
import sys


def per_kernel(db_path, counters):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                      "group by kernel_name, counter_name").fetchall()
    out = {}
    for name, cname, n, val, dur in rows:
        if cname in counters:
            out.setdefault(name.split("(")[0], {})[cname] = (n, val, dur)
    return out


def main(sq_db, fetch_db, write_db, out_md, out_json, config="c2", mode="f32", pairs=32, n_kpts=1024):
    pairs, n_kpts = int(pairs), int(n_kpts)
    sq = per_kernel(sq_db, {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                            "SQ_ACTIVE_INST_ANY"})
    fe = per_kernel(fetch_db, {"FETCH_SIZE"})
    wr = per_kernel(write_db, {"WRITE_SIZE"})
    lines = ["| kernel | launches | avg us (PMC run) | FETCH_SIZE KiB (raw) | HBM read MB (x2) | HBM write MB | MFMA busy % | clock GHz |",
             "|---|---|---|---|---|---|---|---|"]
    js = {"kernels": {}}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, {}).get("FETCH_SIZE", (0, 0, 0))[1])):
        n, f, dur = fe.get(k, {}).get("FETCH_SIZE", (0, 0.0, 0.0))
        _, w, _ = wr.get(k, {}).get("WRITE_SIZE", (0, 0.0, 0.0))
        rd, wt = 2.0 * f * 1024, w * 1024
        busy = clock = ""
        s = sq.get(k, {})
        if "GRBM_GUI_ACTIVE" in s and s["GRBM_GUI_ACTIVE"][1] > 0:
            gui, sdur = s["GRBM_GUI_ACTIVE"][1] / 8.0, s["GRBM_GUI_ACTIVE"][2]  # counter is summed over the 8 XCDs
            clock = f"{gui / sdur:.2f}" if sdur > 20000 else ""
            if "SQ_VALU_MFMA_BUSY_CYCLES" in s and s["SQ_VALU_MFMA_BUSY_CYCLES"][1] > 0:
                busy = f"{100.0 * s['SQ_VALU_MFMA_BUSY_CYCLES'][1] / (gui * 1024):.1f}"  # 256 CUs x 4 SIMDs
        lines.append(f"| {k[-48:]} | {n} | {dur / 1e3:.1f} | {f:.0f} | {rd / 1e6:.1f} | {wt / 1e6:.1f} | {busy} | {clock} |")
        # template variants of one kernel (gemm_p2_kernel<OUT, HAS_R>) share an entry: launch-weighted average; each variant
        # also keeps its own entry under its full name
        short = k.split("::")[-1].split("<")[0].strip()
        full = k.split("::")[-1].strip()
        if full != short:
            js["kernels"][full] = {"read_bytes": rd, "write_bytes": wt, "launches": n}
        e = js["kernels"].get(short)
        if e is None or n == 0:
            js["kernels"].setdefault(short, {"read_bytes": rd, "write_bytes": wt, "launches": n})
        else:
            tot = e["launches"] + n
            e["read_bytes"] = (e["read_bytes"] * e["launches"] + rd * n) / tot
            e["write_bytes"] = (e["write_bytes"] * e["launches"] + wt * n) / tot
            e["launches"] = tot
    text = "\n".join(lines)
    open(out_md, "w").write("# rocprofv3 --pmc passes (separate runs: SQ+GRBM, FETCH_SIZE, WRITE_SIZE), bench.py --steps 2 --warmup 1\n\n"
                            + text + "\n")
    sw = js["kernels"].get("sinkhorn_sweep")
    if sw:  # round 2: the only sweep left on the resident path is the FINAL one, which reads every score exactly once
        known = pairs * n_kpts * n_kpts * 4 + pairs * (2 * n_kpts + 8) * 4
        js["calibration"] = {"kernel": "sinkhorn_sweep<FINAL>", "known_read_bytes": known,
                             "measured_read_bytes_x2": sw["read_bytes"], "ratio": sw["read_bytes"] / known}
    try:
        allj = json.load(open(out_json))
        if "workloads" not in allj:
            allj = {}
    except Exception:
        allj = {}
    allj.setdefault("unit", "HBM bytes per launch (average over the launches of one bench.py run); FETCH_SIZE x 2 per MI355X_MICROARCH.md")
    try:  # the sources these counters were taken on: bench.py reports `traffic` only while csrc/ still hashes to this
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from e2e_multi_view_matching_amd.build import _stamp
        js["csrc_stamp"] = _stamp()
    except Exception:
        js["csrc_stamp"] = None
    allj.setdefault("workloads", {}).setdefault(config, {})[mode] = js
    json.dump(allj, open(out_json, "w"), indent=1)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
