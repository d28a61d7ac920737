import sys, json
for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith("{"):
            d = json.loads(line)
            st = d["stage_ms"]
            print("%-20s %.1f Gb/s %.3f ms/step | part %.3f count %.3f (dedupe %.3f) | asm_dev %.3f outputs %.3f | adj %.3f succ %.3f walk %.3f rank %.3f emit %.3f" % (
                path.split("/")[-1], d["value"], d["ms_per_step"], st.get("partition_kernel", 0), st.get("count_kernel", 0), st.get("count_dedupe_kernel", 0),
                st.get("assemble_device_total_host_clock", 0), st.get("outputs_host_clock", 0),
                st.get("adjacency_kernel", 0), st.get("collapse_succ_split", 0), st.get("collapse_walk", 0), st.get("collapse_rank_device", 0), st.get("collapse_emit", 0)))
            for key in ("two_in_flight", "host_pinned", "host_pinned_two_in_flight", "sharded_one_rank"):
                if key in d and "ms_per_step" in d[key]:
                    print("   %-28s %.1f Gb/s %.3f ms/step (%s)%s" % (key, d[key]["value"], d[key]["ms_per_step"], d[key].get("clock"),
                          (" waits/assemble %s" % d[key].get("host_waits_per_assemble")) if key == "sharded_one_rank" else ""))
                elif key in d:
                    print("   %-28s %s" % (key, d[key]))
            for name, leg in (d.get("legs") or {}).items():
                print("   leg %-32s %.1f Gb/s %.3f ms/step" % (name, leg["value"], leg["ms_per_step"]))
            r = d.get("roofline", {})
            print("   roofline %s frac %.4f count_step_frac %.4f kernel_ms %.3f ; peak_device_bytes %s" % (r.get("kernel"), r.get("frac", 0), r.get("count_step_frac", 0), r.get("kernel_ms", 0), d.get("config", {}).get("peak_device_bytes")))
