import sys, json
for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith("{"):
            d = json.loads(line)
            st = d["stage_ms"]
            print("%-28s %.2f Gb/s  %.1f ms/step | part %.2f count %.2f | asm_dev %.2f outputs %.2f | rank %.2f adj %.2f succ %.2f walk %.2f" % (
                path.split("/")[-1], d["value"], d["ms_per_step"], st.get("partition_kernel", 0), st.get("count_kernel", 0),
                st.get("assemble_device_total_host_clock", 0), st.get("outputs_host_clock", 0), st.get("collapse_host_rank", 0),
                st.get("adjacency_kernel", 0), st.get("collapse_succ_split", 0), st.get("collapse_walk", 0)))
