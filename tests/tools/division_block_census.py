"""Census of the division blocks of the headline forest (tests/tools: it evaluates the trees with the CPU oracle's generator and numpy).
A block = the 512 rows of one datapoint tile at one division node, as the interpreter sees them (csrc/gen/gen_tc_asm.py DIVRANGE):
in range ("fast"), numerator +-0 everywhere ("xzero"), denominator +-0 everywhere ("yzero"), anything else.  Second line: the
operand forms of the divisions the compiler leaves as divisions.   python tests/tools/division_block_census.py > profiles/r03_division_block_census.txt"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.pyoracle import Oracle, depth2leaf, roulette_uniform
o=Oracle("port")
d2l=depth2leaf(6,0.2); rou=roulette_uniform([1,2,3,4]); cs=np.array([-1,0,1],np.float32)
N=4000
v,t,s=o.generate(N,64,10,1,0.0,0.5,[42,0],d2l,rou,cs)
rng=np.random.default_rng(1234)
X=rng.uniform(-5,5,(1024,10)).astype(np.float32)
LO,HI=np.float32(2.0**-46),np.float32(2.0**46)
cnt={}
def add(k): cnt[k]=cnt.get(k,0)+1
forms={}
np.seterr(all='ignore')
for i in range(N):
    n=int(s[i,0]); 
    # evaluate prefix tree recursively (postfix via reverse scan)
    stack=[]
    for j in range(n-1,-1,-1):
        ty=int(t[i,j])&0x7f; val=v[i,j]
        if ty==0: stack.append(('V',X[:,int(val)].copy()))
        elif ty==1: stack.append(('C',np.full(1024,val,np.float32)))
        else:
            f=int(val)
            if ty==2:
                a=stack.pop(); 
                stack.append(('S',np.zeros(1024,np.float32))); continue
            (ka,a)=stack.pop(); (kb,b)=stack.pop()
            if f==1: r=a+b
            elif f==2: r=a-b
            elif f==3: r=a*b
            elif f==4:
                r=np.where(b==0,np.float32(np.nan),a/b)
                if ka=='C' and kb=='C':
                    pass
                else:
                    form=ka+kb
                    if kb=='C':
                        c=b[0]; cb=abs(c)
                        m=np.frexp(cb)[0] if cb>0 else 0
                        if cb==0 or m==0.5: form=None  # becomes mul
                    if form:
                        forms[form]=forms.get(form,0)+2
                        for tile in range(2):
                            x=a[tile*512:(tile+1)*512]; y=b[tile*512:(tile+1)*512]
                            ax,ay=np.abs(x),np.abs(y)
                            okx=np.isnan(x)|((ax>=LO)&(ax<=HI)); oky=np.isnan(y)|((ay>=LO)&(ay<=HI))
                            if okx.all() and oky.all(): add('fast')
                            elif (y==0).all(): add('yzero')
                            elif (x==0).all(): add('xzero')
                            elif (y==0).any(): add('slow_somezero_y')
                            elif (~oky).any(): add('slow_y_range')
                            else:
                                add('slow_x_'+('somezero' if (x==0).any() else 'range'))
            else: r=np.zeros(1024,np.float32)
            k='C' if (ka=='C' and kb=='C') else 'S'
            stack.append((k,r.astype(np.float32)))
tot=sum(cnt.values())
print({k:round(c/tot,4) for k,c in sorted(cnt.items(),key=lambda kv:-kv[1])}, tot/N/2)
ft=sum(forms.values()); print({k:round(c/ft,3) for k,c in forms.items()})
