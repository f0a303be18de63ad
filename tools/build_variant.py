"""Diagnostic: build liblocotouch_env.so variants with extra -D flags into tools/_bin/<name>.so (LOCOTOUCH_AMD_LIB selects one)."""
import os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locotouch_amd import build
name, flags = sys.argv[1], sys.argv[2:]
keep = build.LIB + ".keep"
if os.path.exists(build.LIB):
    shutil.copy2(build.LIB, keep)
try:
    build.build_lib(force=True, extra_flags=flags)
    os.makedirs(os.path.join(build.REPO, "tools", "_bin"), exist_ok=True)
    shutil.copy2(build.LIB, os.path.join(build.REPO, "tools", "_bin", name + ".so"))
finally:  # a variant that does not compile must not leave the product library missing or replaced
    if os.path.exists(keep):
        os.replace(keep, build.LIB)
print("built", name, flags)
