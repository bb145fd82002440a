import sys, subprocess, gzip, struct
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from bamfuzz import Fuzz
import modkit_amd
bam,fa,bed=Fuzz(1000).write('/tmp/fz',bed=True)
modkit_amd.pileup([bam,'/tmp/dev.bed','--no-filtering'])
subprocess.run(['oracle/modkit_oracle','pileup',bam,'/tmp/ora.bed','--no-filtering'],capture_output=True)
a=open('/tmp/dev.bed').read().splitlines(); b=open('/tmp/ora.bed').read().splitlines()
sa=set(a); sb=set(b)
diff=[l for l in b if l not in sa][:6]
print("oracle-only rows:"); print("\n".join(diff))
print("device-only rows:"); print("\n".join([l for l in a if l not in sb][:6]))
pos=int(diff[0].split('\t')[1]); ctg=diff[0].split('\t')[0]
d=gzip.open(bam).read(); o=4; lt=struct.unpack('<i',d[o:o+4])[0]; o+=4+lt; nref=struct.unpack('<i',d[o:o+4])[0]; o+=4
names=[]
for i in range(nref):
    ln=struct.unpack('<i',d[o:o+4])[0]; names.append(d[o+4:o+4+ln-1].decode()); o+=8+ln
while o<len(d):
    bs=struct.unpack('<i',d[o:o+4])[0]; r=d[o+4:o+4+bs]; o+=4+bs
    tid,p,lq,mq,bn,nc,fl,ls=struct.unpack('<iiBBHHHi',r[:20])
    cig=struct.unpack('<%dI'%nc,r[32+lq:32+lq+4*nc]); rl=sum(c>>4 for c in cig if (c&15) in (0,2,3,7,8))
    if names[tid]==ctg and p<=pos<p+rl:
        aux=r[32+lq+4*nc+(ls+1)//2+ls:]
        i=aux.find(b'MMZ'); j=aux.find(b'MmZ'); k=max(i,j)
        mm=aux[k+3:aux.find(b'\0',k)].decode() if k>=0 else None
        print(r[32:32+lq-1].decode(), 'pos',p,'flag',fl,'len',ls,'ncig',nc, 'MM', (mm[:60] if mm else None), 'hdrs', [t.split(',')[0] for t in mm.split(';') if t] if mm else None)
