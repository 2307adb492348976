#!/usr/bin/env python3
"""Static VALU instruction counts of the blocks of rt_path_kernel_stream from the ISA of a -DRT_LEDGER_MARKS build (the
product kernel plus comment markers at the block boundaries; same register allocation, same spills).

The ISA is cut into basic blocks (labels, branches, markers); a query (start marker, stop markers) sums the VALU instructions
of every basic block reachable from the start marker without passing a stop marker -- both sides of a divergent branch count,
as they are issued (a side that NO lane takes is skipped by s_cbranch_execz: rare variants are queried separately and
subtracted).  Prints a JSON dict {region: valu_instructions}.

    python tools/ledger_static.py [file.s]      (default: compiles raytracing_c_amd/csrc/rt_kernels.hip)"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = os.environ.get("RT_LEDGER_KERNEL", "_Z21rt_path_kernel_streamILi16ELb1ELi1ELb1EEv10RT_KParams")


def kernel_asm(path=None):
    if path is None:
        csrc = os.path.join(ROOT, "raytracing_c_amd", "csrc")
        flags = subprocess.check_output(["make", "-s", "-C", csrc, "print-hipflags"], text=True).split()
        flags = [f for f in flags if f not in ("-Wall",)]
        path = os.path.join(tempfile.mkdtemp(), "marks.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DRT_LEDGER_MARKS", "--cuda-device-only", "-S",
                              os.path.join(csrc, "rt_kernels.hip"), "-o", path], stderr=subprocess.DEVNULL)
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start + 1:end]


class Block:
    def __init__(self, name):
        self.name, self.valu, self.salu, self.vmem, self.lds, self.succ, self.marker = name, 0, 0, 0, 0, [], None


def build_cfg(lines):
    blocks, cur = [], Block("entry")
    blocks.append(cur)
    label_of = {}
    pending_fallthrough = True

    def new_block(name):
        nonlocal cur, pending_fallthrough
        b = Block(name)
        if pending_fallthrough:
            cur.succ.append(b)
        blocks.append(b)
        cur = b
        pending_fallthrough = True
        return b

    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            b = new_block(m.group(1))
            label_of[m.group(1)] = b
            continue
        m = re.search(r"; LEDGER_MARK (\w+)", l)
        if m:
            b = new_block("mark:" + m.group(1))
            b.marker = m.group(1)
            continue
        if not l.startswith("\t") or l.startswith("\t.") or l.startswith("\t;"):
            continue
        op = l.split()[0]
        if op.startswith("v_"):
            cur.valu += 1
        elif op.startswith("s_"):
            cur.salu += 1
        elif op.startswith(("global_", "scratch_", "buffer_", "flat_")):
            cur.vmem += 1
        elif op.startswith("ds_"):
            cur.lds += 1
        if op == "s_branch":
            cur.succ.append(l.split()[1])
            pending_fallthrough = False
            new_block("after_branch")
            pending_fallthrough = True
            cur_prev = blocks[-2]
            cur_prev.succ = [s for s in cur_prev.succ if s is not cur]       # no fall-through after an unconditional branch
        elif op.startswith("s_cbranch"):
            cur.succ.append(l.split()[1])
            new_block("after_cbranch")
        elif op == "s_endpgm":
            pending_fallthrough = False
            new_block("after_end")
            blocks[-2].succ = [s for s in blocks[-2].succ if s is not cur]
            pending_fallthrough = True
    for b in blocks:
        b.succ = [label_of[s] if isinstance(s, str) else s for s in b.succ if not isinstance(s, str) or s in label_of]
    return blocks


def query(blocks, start, stops, field="valu"):
    """Sum of `field` over the basic blocks reachable from marker `start` without entering a block that starts with a marker
    in `stops` (or with `start` again).  0 for a marker the kernel instance does not have."""
    first = next((b for b in blocks if b.marker == start), None)
    if first is None:
        return 0
    seen, todo, total = set(), [first], 0
    while todo:
        b = todo.pop()
        if id(b) in seen:
            continue
        seen.add(id(b))
        total += getattr(b, field)
        for s in b.succ:
            if s.marker is not None and (s.marker in stops or s.marker == start):
                continue
            todo.append(s)
    return total


def regions(blocks):
    q = lambda a, stops: query(blocks, a, set(stops))
    r = {}
    for name in ("env", "pstore", "pload", "shade", "accum", "prim", "start", "leaf", "cullmask", "nfull", "nglob"):
        r[name] = q(name + "_begin", [name + "_end"])
    inner_s = ["env", "pstore", "pload", "shade", "accum", "regen", "prim", "start"]
    r["regen_iter"] = q("regen_begin", ["regen_end"])
    r["s_total"] = q("s_begin", ["s_end"])
    r["s_overhead"] = r["s_total"] - sum(q(n + "_begin", [n + "_end"]) for n in inner_s)
    r["s_to_traversal"] = q("s_end", ["round_begin", "tile_begin", "flush_begin", "s_begin"])
    r["round"] = q("round_begin", ["leaf_begin", "node_begin", "s_begin", "flush_begin", "pop_begin"])
    stops_node = ["cullmask_begin", "nfull_begin", "nglob_begin", "nexact_begin", "nfew_begin", "node_entered", "node_end"]
    r["node_head"] = q("node_begin", stops_node)
    r["node_after_cullmask"] = q("cullmask_end", stops_node)
    r["nfull_tail"] = q("nfull_end", ["node_entered", "node_end"])
    r["node_entered_tail"] = q("node_entered", ["node_end"])
    r["nexact"] = q("nexact_begin", ["node_entered", "node_end"])
    r["nfew_1"] = q("nfew_begin", ["nfew_ret1"])
    r["nfew_second_child"] = q("nfew_ret1", ["nfew_two", "nfew_end"])
    r["nfew_2_merge"] = q("nfew_two", ["nfew_ret2", "nfew_end"])
    r["nfew_third_child"] = q("nfew_ret2", ["nfew_four_begin", "nfew_end"])
    r["nfew_fourth_child"] = q("nfew_four_begin", ["nfew_four_end"])
    r["nfew_rank4"] = q("nfew_four_end", ["nfew_end"])
    r["nfew_tail"] = q("nfew_end", ["node_entered", "node_end"])
    r["node_end_to_pop"] = q("node_end", ["pop_begin"])
    r["leaf_end_to_pop"] = q("leaf_end", ["pop_begin"])
    r["pop_loop_entry"] = q("pop_begin", ["pop_iter_begin", "pop_end"])
    r["pop_iter"] = q("pop_iter_begin", ["pop_iter_end"])
    r["pop_up"] = q("pop_up_begin", ["pop_up_end"])
    r["pop_retest"] = q("pop_retest_begin", ["pop_retest_end"])
    r["pop_retest_rare"] = q("pop_rare_begin", ["pop_rare_end"]) + q("pop_glob_begin", ["pop_glob_end"])
    r["pop_iter_back"] = q("pop_iter_end", ["pop_iter_begin", "pop_end"])
    r["pop_end_to_round"] = q("pop_end", ["round_begin"])
    r["tile_setup"] = q("tile_begin", ["tile_end"]) + q("tile_end", ["s_begin"])
    r["flush"] = q("flush_begin", ["tile_begin"])
    return r


if __name__ == "__main__":
    blocks = build_cfg(kernel_asm(sys.argv[1] if len(sys.argv) > 1 else None))
    print(json.dumps(regions(blocks), indent=1))
