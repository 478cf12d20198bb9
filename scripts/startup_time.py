import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import png_write
with tempfile.TemporaryDirectory() as d:
    ncnn_io.write_model(d + "/models", "realesr-animevideov3-x2", synth.make_weights(2))
    os.makedirs(d + "/in"); os.makedirs(d + "/out")
    f = synth.toon_frame(0, 1920, 1080)
    for i in range(3):
        png_write(f"{d}/in/frame{i + 1:08d}.png", f)
    exe = os.path.join(os.getcwd(), "reve_amd", "realesrgan-hip")
    for k in range(4):
        t0 = time.time()
        r = subprocess.run([exe, "-i", d + "/in", "-o", d + "/out", "-s", "2", "-m", d + "/models", "-t", "full"], capture_output=True, text=True, env=dict(os.environ, REVE_DIR_STATS="1", AMD_LOG_LEVEL="0"))
        dt = time.time() - t0
        call = [l for l in r.stderr.splitlines() if "frames in" in l or l.startswith("[main]") or "first frame" in l]
        print(f"run {k}: wall {dt*1e3:.0f} ms; " + "\n   ".join(c[:700] for c in call))
    t0 = time.time(); subprocess.run(["/bin/true"]); print("spawn /bin/true", (time.time()-t0)*1e3, "ms")
    src = "#include <hip/hip_runtime.h>\n#include <cstdio>\nint main(){int n=0; hipGetDeviceCount(&n); void*p; hipMalloc(&p,1<<20); hipFree(p); return 0;}"
    open(d + "/h.cpp", "w").write(src)
    if subprocess.run(["hipcc", "-O2", d + "/h.cpp", "-o", d + "/h"], capture_output=True).returncode == 0:
        for k in range(3):
            t0 = time.time(); subprocess.run([d + "/h"]); print("bare HIP init + 1 MB malloc:", round((time.time()-t0)*1e3), "ms")
