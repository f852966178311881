#!/usr/bin/env python3
"""Approximate VGPR liveness over the ISA of ONE kernel (a backward dataflow pass over the instruction list with its branch targets): where the
register pressure of a hand-scheduled kernel peaks, which values are alive there and where they were defined -- what the allocator's "spilled N"
does not tell.  Written for csrc/mgn_ppr.inc (192 registers of resident weights leave 64 for everything else).
usage: tools/asm_kernels.sh k_edge_bwd_ppr        # -> /tmp/k.s
       python tools/isa_liveness.py /tmp/k.s [print the live set at this instruction index]  [unused]  [region k: carried values]  [index: defs of the live values]
Exec-masked partial writes are treated as full definitions; inline-asm operands are parsed like any instruction."""
import re,sys
src=open(sys.argv[1]).read().split('\n')
ins=[]  # (idx, text)
labels={}
for l in src:
    t=l.split(';')[0].rstrip()
    if not t.strip(): continue
    if re.match(r'^[.\w$]+:',t.strip()) :
        labels[t.strip()[:-1]]=len(ins); continue
    if t.startswith('\t.') or t.strip().startswith('.'): continue
    ins.append(t.strip())
def regs(tok):
    out=set()
    m=re.fullmatch(r'v\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.fullmatch(r'v(\d+)',tok)
    if m: return {int(m.group(1))}
    return out
def parse(t):
    parts=t.split(None,1)
    op=parts[0]; args=[]
    if len(parts)>1:
        args=[x.strip() for x in re.split(r',\s*(?![^\[]*\])',parts[1])]
        # strip modifiers after space
        args=[a.split()[0] if a else a for a in args]
    d=set();u=set()
    nodef=op.startswith(('ds_write','global_store','scratch_store','buffer_store','s_','global_load_lds','v_cmp','v_readlane','v_readfirstlane','ds_bpermute_nodef'))
    if op.startswith('v_cmpx'): nodef=True
    both=('swap' in op)
    acc_use=op.startswith(('v_fmac','v_mac','v_dot2c','v_pk_fmac'))
    for i,a in enumerate(args):
        r=regs(a)
        if not r: continue
        if both: d|=r;u|=r
        elif i==0 and not nodef:
            d|=r
            if acc_use: u|=r
        else: u|=r
    # cndmask etc fine
    return op,d,u
P=[parse(t) for t in ins]
n=len(ins)
succ=[[] for _ in range(n)]
for i,t in enumerate(ins):
    op=t.split()[0]
    if op=='s_endpgm': continue
    if op in('s_branch',) :
        succ[i].append(labels[t.split()[1]]); continue
    if op.startswith('s_cbranch'):
        tgt=t.split()[1]
        if tgt in labels: succ[i].append(labels[tgt])
    if i+1<n: succ[i].append(i+1)
live_in=[set() for _ in range(n)]
changed=True
it=0
while changed:
    changed=False; it+=1
    for i in range(n-1,-1,-1):
        out=set()
        for s in succ[i]: out|=live_in[s]
        op,d,u=P[i]
        new=(out-d)|u
        if new!=live_in[i]:
            live_in[i]=new; changed=True
mx=max(range(n),key=lambda i:len(live_in[i]))
print('iters',it,'max live',len(live_in[mx]),'at',mx,ins[mx])
# report per barrier region max
reg=0;cur=0;curi=0
for i,t in enumerate(ins):
    if t.startswith('s_barrier'):
        print('region',reg,'ends at',i,'max live',cur,'at',curi,ins[curi][:70]); reg+=1;cur=0
    if len(live_in[i])>cur: cur=len(live_in[i]);curi=i
print('last region max',cur,ins[curi][:70])
if len(sys.argv)>2:
    i=int(sys.argv[2]); 
    s=sorted(live_in[i]); print(s)
# live-through analysis for region 3: registers live at every point of the region and never referenced in it
bars=[i for i,t in enumerate(ins) if t.startswith('s_barrier')]
def region(k): return range(bars[k-1]+1 if k>0 else 0, bars[k])
for k in (3,5):
    rg=list(region(k))
    touched=set()
    for i in rg:
        op,d,u=P[i]; touched|=d|u
    through=set(live_in[rg[0]])
    for i in rg: through&=live_in[i]
    lt=sorted(through-touched)
    print('region',k,'live-through untouched',len(lt),lt)
    # where are they next used?
    for v in lt:
        # find next use after region
        nxt=None
        for i in range(rg[-1],n):
            op,d,u=P[i]
            if v in u: nxt=(i,ins[i][:80]); break
        print('  v%d next use:'%v,nxt)
if len(sys.argv)>3:
    k=int(sys.argv[3])
    rg=list(region(k))
    touched=set()
    for i in rg:
        op,d,u=P[i]; touched|=d|u
    through=set(live_in[rg[0]])
    for i in rg: through&=live_in[i]
    print('REGION',k,'touched',len(touched),'live-through untouched',len(through-touched))
    # touched registers that are live at region start AND region end (carried values that are used)
    carried=[v for v in sorted(touched) if v in live_in[rg[0]] and v in live_in[rg[-1]]]
    print(' carried+touched',len(carried),carried)
    print(' untouched-through (non-W guess >=194 or small):',[v for v in sorted(through-touched)])
if len(sys.argv)>4:
    i=int(sys.argv[4])
    L=sorted(live_in[i])
    # find W regs = those untouched in whole loop body? approximate: regs whose last def is before first barrier
    firstbar=bars[0]
    for v in L:
        # last def before i
        dd=None
        for j in range(i-1,-1,-1):
            if v in P[j][1]: dd=j; break
        if dd is not None and dd>firstbar:
            # next use
            nu=None
            for j in range(i,n):
                if v in P[j][2]: nu=j; break
            print('v%d def@%d %s | next use@%s %s'%(v,dd,ins[dd][:60],nu,ins[nu][:50] if nu else ''))
