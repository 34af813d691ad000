"""Runs tests/emu/_build/emu_cli (host emulation of the device code under ASan/UBSan) — TEST INFRASTRUCTURE."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tests", "emu", "_build", "emu_cli")


def ensure():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    return CLI


def dump_many(sources, what, env=None):
    """[(status, text)] for every source (bytes); env: extra environment of the emulator (its EMU_* switches)."""
    ensure()
    out = []
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for i, s in enumerate(sources):
            p = os.path.join(d, f"{i}.opus")
            with open(p, "wb") as f:
                f.write(s)
            paths.append(p)
        for lo in range(0, len(paths), 200):
            r = subprocess.run([CLI, str(what)] + paths[lo:lo + 200], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env) if env else None)
            if r.returncode != 0:
                raise RuntimeError(f"emu_cli failed (sanitizer?) rc={r.returncode}\n{r.stderr.decode(errors='replace')[-4000:]}")
            data = r.stdout
            pos = 0
            while pos < len(data):
                assert data.startswith(b"== ", pos)
                eol = data.index(b"\n", pos)
                head = data[pos:eol].decode().split(" ")
                status, n = int(head[-3]), int(head[-1])
                out.append((status, data[eol + 1:eol + 1 + n].decode("utf-8", errors="surrogateescape")))
                pos = eol + 1 + n
    return out


def dump(src, what):
    return dump_many([src], what)[0]
