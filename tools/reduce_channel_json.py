"""rocprofv3 --output-format json counter records -> per-channel statistics of k_entity_stream (tools/placement_channels.sh).
The JSON layout is discovered, not assumed: counter records are found as dicts holding a numeric value next to a counter / dimension id;
the first run prints what it found to stderr."""
import collections, glob, json, sys

root, out = sys.argv[1], sys.argv[2]
files = glob.glob(root + "/**/*results.json", recursive=True) + glob.glob(root + "/**/*.json", recursive=True)
doc = json.load(open(files[0]))
top = doc["rocprofiler-sdk-tool"][0] if "rocprofiler-sdk-tool" in doc else doc
print("top keys", list(top.keys())[:20], file=sys.stderr)
# kernel symbols: id -> name
names = {}
for k in top.get("kernel_symbols", []):
    names[k.get("kernel_id")] = k.get("formatted_kernel_name") or k.get("kernel_name") or ""
# counter ids -> name, and dimension decode
cinfo = {}
for agent_counters in (top.get("counters") or []):
    cid = agent_counters.get("id", {}).get("handle") if isinstance(agent_counters.get("id"), dict) else agent_counters.get("id")
    cinfo[cid] = agent_counters.get("name")
print("counters known", len(cinfo), list(cinfo.items())[:3], file=sys.stderr)
recs = (top.get("callback_records") or {}).get("counter_collection") or top.get("counter_collection") or []
print("records", len(recs), "first", json.dumps(recs[0])[:900] if recs else None, file=sys.stderr)
per = collections.defaultdict(lambda: collections.defaultdict(list))   # counter name -> instance key -> values (one per dispatch)
dur = []
for r in recs:
    info = r.get("dispatch_data", {}).get("dispatch_info", {})
    kname = names.get(info.get("kernel_id"), "")
    if "k_entity_stream" not in kname:
        continue
    if "start_timestamp" in r.get("dispatch_data", {}):
        dur.append((r["dispatch_data"]["end_timestamp"] - r["dispatch_data"]["start_timestamp"]) * 1e-6)
    seen = collections.Counter()
    for rec in r.get("records", []):
        cid = rec.get("counter_id", {}).get("handle") if isinstance(rec.get("counter_id"), dict) else rec.get("counter_id")
        inst = seen[cid]            # a counter's records of one dispatch come one per instance (16 channels x 8 XCDs), in a fixed order
        seen[cid] += 1
        per[cinfo.get(cid, str(cid))][inst].append(float(rec.get("value", rec.get("counter_value", 0.0))))
summary = {"stream_ms": sum(dur) / len(dur) if dur else None, "launches": len(dur)}
line = f"{root.split('/')[-1]:8s} stream {summary['stream_ms'] if dur else float('nan'):.3f} ms"
for cname, inst in per.items():
    vals = sorted(sum(v) / len(v) for v in inst.values())
    if not vals:
        continue
    mean = sum(vals) / len(vals)
    sd = (sum((x - mean) ** 2 for x in vals) / len(vals)) ** 0.5
    by_inst = [sum(v) / len(v) for _k, v in sorted(inst.items())]
    xcd = [sum(by_inst[i:i + 16]) for i in range(0, len(by_inst), 16)] if len(by_inst) % 16 == 0 else []
    summary[cname] = {"instances": len(vals), "mean": mean, "min": vals[0], "max": vals[-1], "cv": sd / mean if mean else None,
                      "per_instance": by_inst, "per_group_of_16": xcd}
    line += f" | {cname}: {len(vals)} instances, max/mean {vals[-1] / mean:.3f}, min/mean {vals[0] / mean:.3f}, cv {sd / mean:.4f}"
    if xcd:
        m = sum(xcd) / len(xcd)
        line += f", groups of 16 max/mean {max(xcd) / m:.3f}"
json.dump(summary, open(out, "w"), indent=1)
print(line)
