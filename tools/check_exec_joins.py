"""Static guard against the hipcc 7.0 miscompile found in round 6 (profiles/r06_persist_spw67_fault.txt): at the join of an
exec-masked region (the target of `s_cbranch_execz` after `s_and_saveexec_b64 sX, ...`), every lane-wise instruction must
come AFTER `s_or_b64 exec, exec, sX`.  In k_pcg_persist<3,6,1> / <3,7,0> the register allocator's copies of long-lived
values into accumulation registers (`v_accvgpr_write_b32 aN, vM`) were placed BEFORE the restore: lanes outside the mask
(padding lanes) kept garbage, which was used under full exec much later.

usage: python tools/check_exec_joins.py <file.s> [kernel-name-substring]  -> lists offending joins, exit 1 if any"""
import re
import sys


def kernels(text):
    for m in re.finditer(r"^(_Z\w+):", text, re.M):
        end = text.find(".Lfunc_end", m.end())
        yield m.group(1), text[m.end():end].split("\n")


def check(lines):
    labels = {}
    for k, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    bad = []
    for k, l in enumerate(lines):
        m = re.match(r"\s+s_cbranch_execz\s+(\.LBB\d+_\d+)", l)
        if not m or m.group(1) not in labels:
            continue
        # the saveexec that opened the region: the closest one above
        save = None
        for q in range(k - 1, max(k - 6, -1), -1):
            ms = re.match(r"\s+s_and_saveexec_b64\s+(s\[\d+:\d+\])", lines[q])
            if ms:
                save = ms.group(1)
                break
        if save is None:
            continue
        j = labels[m.group(1)] + 1
        pending = []
        while j < len(lines):
            t = lines[j].strip()
            j += 1
            if not t or t.startswith((";", ".")):
                if re.match(r"^\.LBB", t):
                    break
                continue
            if re.match(r"s_or_b64\s+exec,\s*exec,\s*" + re.escape(save), t):
                break
            if "exec" in t and t.startswith("s_"):
                break                                          # another exec manipulation: not the simple join pattern
            op = t.split()[0]
            lanewise = op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")) and not op.startswith(("v_writelane", "v_readlane", "v_readfirstlane"))
            if lanewise:
                pending.append(t)
            if op.startswith(("s_branch", "s_cbranch", "s_endpgm")):
                break
        else:
            pending = []
        if pending and j < len(lines):
            bad.append((m.group(1), pending[:4]))
    return bad


def main():
    text = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    total = 0
    for name, lines in kernels(text):
        if pat not in name:
            continue
        bad = check(lines)
        if bad:
            total += len(bad)
            print(f"{name[:110]}: {len(bad)} join(s) with lane-wise instructions before the exec restore")
            for lab, ins in bad[:3]:
                print("   ", lab, "|", "; ".join(ins))
    print("clean" if not total else f"{total} offending joins")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
