import numpy as np, ctypes
libm = ctypes.CDLL('libm.so.6'); libm.asinf.restype = ctypes.c_float; libm.asinf.argtypes = [ctypes.c_float]
f32 = np.float32
one=f32(1.0); pio2_hi=f32(1.57079637050628662109375); pio2_lo=f32(-4.37113900018624283e-8); pio4_hi=f32(0.785398185253143310546875)
p0=f32(1.666675248e-1); p1=f32(7.495297643e-2); p2=f32(4.547037598e-2); p3=f32(2.417951451e-2); p4=f32(4.216630880e-2)
def asinf(x):
    x=f32(x); hx=np.frombuffer(x.tobytes(),dtype=np.int32)[0]; ix=hx&0x7fffffff
    if ix==0x3f800000: return f32(x*pio2_hi+x*pio2_lo)
    if ix>0x3f800000: return f32(np.nan)
    if ix<0x3f000000:
        if ix<0x32000000: return x
        t=f32(x*x); w=f32(t*f32(p0+f32(t*f32(p1+f32(t*f32(p2+f32(t*f32(p3+f32(t*p4)))))))))
        return f32(x+f32(x*w))
    w=f32(one-abs(x)); t=f32(w*f32(0.5))
    p=f32(t*f32(p0+f32(t*f32(p1+f32(t*f32(p2+f32(t*f32(p3+f32(t*p4)))))))))
    s=f32(np.sqrt(t))
    if ix>=0x3F79999A:
        t=f32(pio2_hi-f32(f32(f32(2.0)*f32(s+f32(s*p)))-pio2_lo))
    else:
        iw=np.frombuffer(s.tobytes(),dtype=np.int32)[0]
        w=np.frombuffer(np.int32(iw&np.int32(-4096)).tobytes(),dtype=np.float32)[0]
        c=f32(f32(t-f32(w*w))/f32(s+w)); r=p
        p=f32(f32(f32(f32(2.0)*s)*r)-f32(pio2_lo-f32(f32(2.0)*c)))
        q=f32(pio4_hi-f32(f32(2.0)*w))
        t=f32(pio4_hi-f32(p-q))
    return t if hx>0 else f32(-t)
rng=np.random.default_rng(0); bad=0; N=300000
xs=np.concatenate([rng.uniform(-1,1,N), rng.uniform(-0.5,0.5,N//2), np.array([0.5,-0.5,0.975,0.9749999,1.0,-1.0,1e-9,0.0])]).astype(np.float32)
for x in xs:
    a=asinf(x); b=f32(libm.asinf(float(x)))
    if a.tobytes()!=b.tobytes():
        bad+=1
        if bad<6: print('mismatch',x,a,b)
print('mismatches',bad,'of',len(xs))
