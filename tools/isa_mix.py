"""Static instruction mix of one kernel of an ISA listing (make -C hmvec_amd/csrc asm).
Usage: python tools/isa_mix.py hmvec_amd/csrc/longgrid.s <mangled-name-substring> [top-n]"""
import collections, sys
path, pat = sys.argv[1], sys.argv[2]
topn = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lines = open(path).read().split("\n")
starts = [i for i, l in enumerate(lines) if l.endswith(":") is False and pat in l and l.split(":")[0].startswith("_Z") and ":" in l and not l.startswith("\t")]
for s in starts:
    name = lines[s].split(":")[0]
    e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cnt = collections.Counter()
    for l in lines[s + 1:e]:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        cnt[l.split()[0]] += 1
    grp = collections.Counter()
    for k, v in cnt.items():
        g = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_")
             else "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
        grp[g] += v
    print(name, sum(cnt.values()), dict(grp))
    print("  scratch:", {k: v for k, v in cnt.items() if k.startswith("scratch_")})
    print("  " + ", ".join(f"{k} {v}" for k, v in cnt.most_common(topn)))
