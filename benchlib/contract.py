"""The contract line of bench.py: ONE small JSON object, last on stdout.

The driver keeps only the last 8 000 bytes of stdout (round 5: a 26 KB line lost its head and the run went unmeasured), so
the line that goes to stdout is the contract head -- metric, value, unit, n_gpus, steps, warmup, ms_per_step, dtype, config,
roofline, cpu_baseline, parity -- plus one compact {value, unit, frac, cpu, parity} tuple per secondary.  Everything else
(the full objects bench.py has always assembled) goes to a side file, `bench_secondary.json`, and to a short table on stderr.

compact_line() never drops the head: when the line would pass LIMIT it sheds, in this order, the per-secondary tuples'
optional members, then whole secondaries (largest first), then the optional head members (`end_to_end`, `exchange`, ...).
"""
import json

LIMIT = 6000           # bytes of the stdout line incl. newline (the driver's window is 8 000: keep slack for a banner before it)
HEAD = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
STR_CAP = 160          # strings inside the line are cut to this many characters (full text: the side file)


def _cut(s, cap=STR_CAP):
    return s if len(s) <= cap else s[:cap - 3] + "..."


def _num(x, digits=6):
    """floats at 6 significant digits (the side file keeps them in full)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float("%.*g" % (digits, x))


def _slim(o, cap=STR_CAP):
    if isinstance(o, dict):
        return {k: _slim(v, cap) for k, v in o.items() if v is not None or k in ("traffic", "vs_baseline")}
    if isinstance(o, (list, tuple)):
        return [_slim(v, cap) for v in o]
    if isinstance(o, str):
        return _cut(o, cap)
    return _num(o)


def _roofline(r, cap=STR_CAP):
    if not isinstance(r, dict):
        return r
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "launches", "algorithmic_bytes_per_launch",
            "algorithmic_bytes_per_instance", "traffic_source")
    out = {k: r[k] for k in keep if k in r}
    out.setdefault("traffic", None)
    return _slim(out, cap)


def _cpu(c, cap=STR_CAP):
    if not isinstance(c, dict):
        return c
    return _slim({k: c[k] for k in ("value", "unit", "cores", "kind", "sample") if k in c}, cap)


def _parity(p):
    if not isinstance(p, dict):
        return p
    for k in ("bit_exact", "identical", "ok"):
        if k in p:
            return bool(p[k])
    return _slim(p, 60)


def _tuple(r):
    """one secondary -> {value, unit, frac, cpu, parity} (+ ms_per_step, traffic_ratio, quality deltas when the object has them)"""
    if not isinstance(r, dict):
        return None
    if "error" in r and "value" not in r:
        return {"error": _cut(str(r["error"]), 80)}
    t = {}
    for k in ("value", "unit", "ms_per_step"):
        if r.get(k) is not None:
            t[k] = _num(r[k])
    rf = r.get("roofline")
    if isinstance(rf, dict):
        if rf.get("frac") is not None:
            t["frac"] = _num(rf["frac"], 4)
        if rf.get("traffic") and rf.get("algorithmic_bytes_per_launch"):
            t["traffic_ratio"] = _num(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 3)
    cb = r.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value") is not None:
        t["cpu"] = _num(cb["value"])
    if r.get("parity") is not None:
        t["parity"] = _parity(r["parity"])
    for k in ("rmse_minus_sequential", "rmse_test_after_run", "pair_accuracy_test_after_run"):
        if r.get(k) is not None:
            t[k] = _num(r[k])
    db = r.get("dag_bound")
    if isinstance(db, dict):
        ob = db.get("measured_over_chain_bound", db.get("measured_over_bound"))
        if ob is not None:
            t["over_dag_bound"] = _num(ob, 3)
    cd = r.get("contract_delta_vs_exact")
    if isinstance(cd, dict):
        t["delta_vs_exact"] = _slim(cd)
    return t or None


def _orders(o):
    """secondary.orders: {stream: {exact, window, auto, cpu, parity}} -> flat tuples"""
    out = {}
    for name, s in o.items():
        if not isinstance(s, dict):
            continue
        for mode in ("exact", "window"):
            m = s.get(mode)
            if isinstance(m, dict):
                t = _tuple(dict(m, cpu_baseline=s.get("cpu_baseline"), parity=s.get("parity") if mode == "exact" else None))
                if t:
                    out["%s.%s" % (name, mode)] = t
        au = s.get("auto")
        if isinstance(au, dict) and au.get("decision") is not None:
            out["%s.auto" % name] = {"decision": au["decision"], "value": _num(au.get("value"))}
        dflt = s.get("default")
        if isinstance(dflt, dict):
            t = _tuple(dict(dflt, cpu_baseline=s.get("cpu_baseline")))
            if t:
                if dflt.get("path"):
                    t["path"] = _cut(str(dflt["path"]), 60)
                out["%s.default" % name] = t
    return out


def secondary_tuples(secondary):
    out = {}
    for name, r in (secondary or {}).items():
        if name == "orders" and isinstance(r, dict):
            out.update({"orders." + k: v for k, v in _orders(r).items()})
        elif name == "single_process_handle" and isinstance(r, dict):
            for xch, rr in r.items():
                t = _tuple(rr)
                if t:
                    out["single_process_handle.%s" % xch] = t
        elif isinstance(r, dict):
            t = _tuple(r)
            if t:
                out[name] = t
        elif isinstance(r, str):
            out[name] = _cut(r, 80)
    return out


def _exchange(x):
    if not isinstance(x, dict):
        return x
    keep = ("step", "backend", "transport", "windows", "bytes_per_window", "world_size_reported", "ladder_rung", "ladder", "fallback")
    return _slim({k: x[k] for k in keep if x.get(k) is not None}, 100)


def compact_line(full, limit=LIMIT):
    """full = the object bench.py assembled (any size) -> (line, dropped): line is a JSON string of at most `limit` - 1 bytes holding
    the contract head and the compact secondaries; dropped lists what had to be shed to fit."""
    line = {k: full[k] for k in HEAD if k in full}
    line["value"] = full.get("value")
    cfg = full.get("config") or {}
    line["config"] = _slim(cfg, 120)
    line["roofline"] = _roofline(full.get("roofline"))
    line["cpu_baseline"] = _cpu(full.get("cpu_baseline"))
    line["parity"] = _slim(full.get("parity"), 120)
    optional = []          # (key, value) shed last-to-first when the line is too long
    for k in ("rmse_test_after_run", "passes_before_rmse", "rmse_sequential_reference", "rmse_minus_sequential", "pair_accuracy_test_after_run"):
        if full.get(k) is not None:
            line[k] = _num(full[k], 8)
    if full.get("exchange"):
        optional.append(("exchange", _exchange(full["exchange"])))
    ar = full.get("allreduce_step")
    if isinstance(ar, dict):
        keep = ("value", "unit", "ms_per_step", "rmse_minus_sequential", "backend", "windows", "bytes_per_window", "n_gpus", "measured_as", "error")
        optional.append(("allreduce_step", _slim({k: ar[k] for k in keep if ar.get(k) is not None}, 100)))
    if isinstance(full.get("roofline_aggregate"), dict):
        optional.append(("roofline_aggregate", _roofline(full["roofline_aggregate"], 80)))
    if isinstance(full.get("end_to_end"), dict):
        optional.append(("end_to_end", _slim({k: full["end_to_end"][k] for k in ("rounds", "value", "unit") if k in full["end_to_end"]})))
    if isinstance(full.get("launch_model"), dict):
        lm = full["launch_model"]
        optional.append(("launch_model", _slim({k: lm[k] for k in ("model_us", "measured_us", "measured_over_model") if k in lm})))
    if full.get("secondary_error"):
        optional.append(("secondary_error", _cut(str(full["secondary_error"]), 200)))
    for k, v in optional:
        line[k] = v
    sec = secondary_tuples(full.get("secondary"))
    if sec:
        line["secondary"] = sec
    line["details"] = full.get("details", "bench_secondary.json")

    dropped = []
    limit -= 240          # room for the note of what was shed

    def size():
        return len(json.dumps(line, separators=(",", ":"))) + 1

    # 1: long strings in the head
    if size() > limit:
        for obj, key in ((line["roofline"], "traffic_source"), (line["roofline"], "kernel"), (line["cpu_baseline"], "sample"), (line["config"], "workload")):
            if isinstance(obj, dict) and isinstance(obj.get(key), str) and len(obj[key]) > 60:
                obj[key] = _cut(obj[key], 60)
                dropped.append("cut:" + key)
    # 2: the optional members of the secondary tuples
    if size() > limit and sec:
        for t in sec.values():
            if isinstance(t, dict):
                for k in ("delta_vs_exact", "over_dag_bound", "traffic_ratio", "ms_per_step", "rmse_test_after_run", "pair_accuracy_test_after_run", "unit"):
                    t.pop(k, None)
        dropped.append("secondary:optional-members")
    # 3: whole secondaries, largest first
    while size() > limit and sec:
        k = max(sec, key=lambda n: len(json.dumps(sec[n])))
        sec.pop(k)
        dropped.append("secondary:" + k)
    if not sec:
        line.pop("secondary", None)
    # 4: the optional head members, last added first
    for k, _ in reversed(optional):
        if size() <= limit:
            break
        line.pop(k, None)
        dropped.append(k)
    if dropped:
        limit += 240
        line["dropped_to_fit"] = dropped if len(json.dumps(dropped)) <= 200 else dropped[:3] + ["... %d in all (see the side file)" % len(dropped)]
    else:
        limit += 240
    s = json.dumps(line, separators=(",", ":"))
    assert len(s) + 1 <= limit, "contract line is %d bytes, over the %d-byte limit even after shedding everything optional" % (len(s) + 1, limit)
    assert "\n" not in s
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "roofline", "cpu_baseline"):
        assert k in line, "contract line lost %r" % k
    return s, dropped


def stderr_table(full):
    """the secondaries as one short line each (what used to be readable only inside the 26 KB object)"""
    rows = []
    for name, t in secondary_tuples(full.get("secondary")).items():
        if isinstance(t, dict):
            rows.append("  %-44s %s" % (name, " ".join("%s=%s" % (k, json.dumps(v)) for k, v in t.items())))
        else:
            rows.append("  %-44s %s" % (name, t))
    return "\n".join(rows)
