import sys, subprocess
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import navtex_amd as nv, signals
st,_=signals.stream_params(nv, 778, nv.RATE_IN, n_phasing=12, text="ZCZC RT42\nREAL TIME 5678\nNNNN\n")
iq=nv.synth_host(st, nv.RATE_IN, 26*nv.FRAME_IN); iq.tofile('/tmp/iq.bin')
subprocess.run(["gcc","-O2","tests/harness/shim_realtime.c","-o","/tmp/shim_rt","-Lnavtex_amd","-lnavtex_amd","-Wl,-rpath,"+"navtex_amd","-Wl,-rpath,/opt/rocm/lib"],check=True)
out=subprocess.run(["/tmp/shim_rt","/tmp/iq.bin"],capture_output=True,text=True).stdout
print("\n".join(l for l in out.splitlines() if l.startswith(("LAT","MSG","BURST"))))
fed=[l for l in out.splitlines() if l.startswith("FED")]
print(fed[-3:])
