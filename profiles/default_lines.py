"""Lines of icp_rust_amd/csrc that the DEFAULT build (libicp_mi355x.so: none of the development macros defined) compiles,
next to the lines of the files: what `make experiments` / the profiling variants add is behind #ifdef.  VERDICT r4 item 7
asked for the default library to shed device code; this is the count that says by how much.
usage: python profiles/default_lines.py [git-rev]   (a revision: count that tree instead of the working copy)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = "icp_rust_amd/csrc"
DEV = {"ICP_EXPERIMENTS", "ICP_NN_STATS", "ICP_COOP_PROFILE", "ICP_WIN_DEBUG", "ICP_LOOP_PROFILE", "ICP_TILE_PROFILE",
       "ICP_HACK_CAP", "ICP_TINY_PROFILE", "ICP_QS_DEBUG", "ICP_FAST_PROFILE", "ICP_PICK_PROFILE"}


def default_lines(text):
    """lines outside blocks that only a development macro switches on (#ifdef DEV ... [#else kept] ... #endif)"""
    kept, stack = 0, []  # stack of (is_dev_block, currently_excluded)
    for ln in text.splitlines():
        s = ln.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b\s*(.*)", s)
        if m:
            kind, rest = m.group(1), m.group(2)
            if kind in ("ifdef", "ifndef", "if"):
                name = rest.split()[0] if rest.split() else ""
                name = re.sub(r"^defined\(?|\)$", "", name)
                dev = name in DEV
                stack.append([dev, dev and kind != "ifndef"])
            elif kind in ("else", "elif") and stack:
                if stack[-1][0]:
                    stack[-1][1] = not stack[-1][1]
            elif kind == "endif" and stack:
                stack.pop()
            if not any(ex for _, ex in stack) and not (stack and stack[-1][0]) and not (kind == "endif"):
                kept += 1
            continue
        if not any(ex for _, ex in stack):
            kept += 1
    return kept


def main():
    rev = sys.argv[1] if len(sys.argv) > 1 else None
    if rev:
        names = subprocess.check_output(["git", "ls-tree", "-r", "--name-only", rev, CSRC], cwd=ROOT, text=True).split()
        read = lambda p: subprocess.check_output(["git", "show", f"{rev}:{p}"], cwd=ROOT, text=True)
    else:
        names = [f"{CSRC}/{f}" for f in sorted(os.listdir(os.path.join(ROOT, CSRC)))]
        read = lambda p: open(os.path.join(ROOT, p)).read()
    tot_f = tot_d = 0
    for p in names:
        if not p.endswith((".hip", ".hpp")):
            continue
        t = read(p)
        f, d = len(t.splitlines()), default_lines(t)
        tot_f, tot_d = tot_f + f, tot_d + d
        print(f"{os.path.basename(p):24s} file {f:6d}   default build {d:6d}")
    print(f"{'total':24s} file {tot_f:6d}   default build {tot_d:6d}")


if __name__ == "__main__":
    main()
