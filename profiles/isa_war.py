"""Write-after-read distances of matrix-instruction sources in hipcc -save-temps assembly (wait states between a v_mfma and the next write of a
register it reads as SrcC / SrcA / SrcB): python profiles/isa_war.py <asm> <kernel regex>.  Used for profiles/r04_isa_packed_diff.txt."""
import re, sys, collections
sys.path.insert(0,'profiles')
from isa_hazards import kernels, parse, regs
def war(body, window=24):
    ins=[p for p in (parse(l) for l in body) if p]
    out=[]
    for i,(op,ops) in enumerate(ins):
        if not op.startswith('v_mfma'): continue
        d=regs(ops[0]); a=regs(ops[1]); b=regs(ops[2]); c=regs(ops[3]) if len(ops)>3 else set()
        for which,src in (('C',c-d),('A',a-d),('B',b-d)):
            if not src: continue
            waits=0
            for j in range(i+1,min(i+1+window,len(ins))):
                op2,ops2=ins[j]
                if op2.startswith(('v_','ds_read','global_load','buffer_load')) and ops2:
                    w=regs(ops2[0]) if not op2.startswith(('v_cmp','v_cmpx')) else set()
                    if w & src:
                        out.append((which, waits, i, j, op, op2)); break
                # wait states contributed by this instruction
                if op2=='s_nop': waits+=int(ops2[0])+1
                else: waits+=1
    return out
path=sys.argv[1]; pat=sys.argv[2]
for name,body in kernels(path):
    if not re.search(pat,name): continue
    r=war(body)
    c=collections.Counter((w,d) for w,d,_,_,_,_ in r)
    print(name[:140])
    for which in 'CAB':
        ds=sorted((d,n) for (w,d),n in c.items() if w==which)
        print('   WAR on Src'+which+': (intervening wait states: count)', ds[:12])
