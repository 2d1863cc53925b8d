"""Which packed-fp32 instruction each probe kernel of tools/micro/pk_forms.hip really contains (hipcc -S): the table the tool
prints is only about the forms its source was MEANT to produce if this script agrees.  python tools/micro/pk_forms_check.py"""
import os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tmp = tempfile.mkdtemp()
s = os.path.join(tmp, "pk_forms.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", os.path.join(R, "tools/micro/pk_forms.hip"), "-o", s],
                      stderr=subprocess.DEVNULL)
asm = open(s).read()
for m in re.finditer(r"^(_Z7k_formsILi(\d+)ELb1EEvPKfPji):(.*?)\.Lfunc_end", asm, re.S | re.M):
    body = m.group(3)
    forms = sorted(set(re.sub(r"\s+", " ", re.sub(r"v\[\d+:\d+\]|s\[\d+:\d+\]", "R", ln.split(";")[0].strip())) for ln in body.splitlines() if re.search(r"\bv_pk_\w+_f32\b", ln)))
    print(f"form {int(m.group(2)):2d}:", " || ".join(forms) if forms else "(no packed fp32 instruction!)")
