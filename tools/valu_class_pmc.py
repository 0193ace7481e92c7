"""Which hardware class counter counts which VALU opcode: the single-opcode kernels of tools/valu_calib (k_<name>: 64 instructions of the opcode per loop trip)
under the SQ_INSTS_VALU_* class counters (tools/valu_class_pmc.sh).  Prints JSON: opcode kernel -> {class counter: counted instructions / SQ_INSTS_VALU}.
A class that holds (nearly) all of a kernel's VALU instructions is the opcode's class; the loop's own few instructions show as a per cent or two elsewhere."""
import csv, glob, json, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].strip()
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
out = {}
for k, cs in sorted(acc.items()):
    tot = cs.get("SQ_INSTS_VALU", 0.0)
    if tot <= 0:
        continue
    name = k.replace("void ", "").replace("k_", "", 1)
    out[name] = {c.replace("SQ_INSTS_VALU_", ""): round(v / tot, 4) for c, v in sorted(cs.items()) if c.startswith("SQ_INSTS_VALU_") and v / tot > 0.004}
json.dump(out, sys.stdout, indent=1)
print()
