import re, collections
txt = open("/tmp/stacks.txt").read()
dumps = txt.split("Timeout (")[1:]
print(len(dumps), "dumps")
# for each dump: for each thread, the innermost frame inside svision_amd or bench (file:line func)
for i, d in enumerate(dumps):
    threads = re.split(r"\n(?=Thread 0x|Current thread 0x)", d)
    out = []
    for t in threads[1:]:
        frames = re.findall(r'File "([^"]+)", line (\d+) in (\S+)', t)
        if not frames: continue
        top = frames[0]
        own = next(((f.split("/")[-1], l, fn) for f, l, fn in frames if "svision_amd" in f or f.endswith("bench.py")), None)
        out.append("%s:%s %s | %s" % (top[0].split("/")[-1], top[1], top[2], "%s:%s %s" % own if own else "-"))
    print(i, " || ".join(out))
