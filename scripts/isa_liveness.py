"""VGPR liveness over the control-flow graph of one kernel in a hipcc -save-temps .s file (gfx950): per-instruction live register count by
backward dataflow to a fixed point, the maximum and where it sits, and the live sets at the barriers.  Written to find why a kernel
spilled at a nominal pressure well under 256 (values defined under one `if` and used under another count as live around the whole loop).

    python scripts/isa_liveness.py file.s [kernel-name-substring] [print-threshold]
"""
import re,sys
path=sys.argv[1]; kpat=sys.argv[2] if len(sys.argv)>2 else 'relattn_bwd_fused_kernel'
thr=int(sys.argv[3]) if len(sys.argv)>3 else 0
L=open(path).read().split('\n')
start=[i for i,l in enumerate(L) if re.match(r'^_Z.*'+kpat+r'.*:',l)][0]
end=[i for i,l in enumerate(L) if i>start and '.end_amdhsa_kernel' in l][0]
def regs(tok):
    out=[]
    for m in re.finditer(r'v\[(\d+):(\d+)\]|(?<![a-z_\d\[])v(\d+)(?![\d\]:])',tok):
        if m.group(1): out+=list(range(int(m.group(1)),int(m.group(2))+1))
        else: out.append(int(m.group(3)))
    return out
# instructions
ins=[]  # (line, op, defs, uses, text)
labels={}
for i in range(start+1,end):
    l=L[i]
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=len(ins); continue
    t=l.strip()
    if not l.startswith('\t') or not t or t[0] in '.;': continue
    t=t.split(';')[0].strip()
    if not t: continue
    op=t.split()[0]
    ops=t[len(op):].split(',')
    nodst=op.startswith(('ds_write','buffer_store','scratch_store','global_store','buffer_atomic','global_atomic','s_','ds_add','buffer_wbl2','buffer_inv'))
    if 'lds' in t.split() and op.startswith('buffer_load'): nodst=True
    if nodst: d=[]; u=[r for o in ops for r in regs(o)]
    else:
        d=regs(ops[0]) if ops else []
        u=[r for o in ops[1:] for r in regs(o)]
        if op.startswith(('v_writelane','ds_read_u16_d16','v_mov_b32_dpp','v_fmac','v_pk_fmac','v_dot2c')) : u=u+d
    ins.append((i+1,op,set(d),set(u),t))
n=len(ins)
succ=[[] for _ in range(n)]
for k,(ln,op,d,u,t) in enumerate(ins):
    if op=='s_endpgm': continue
    m=re.search(r'(\.LBB\d+_\d+)',t) if op.startswith(('s_cbranch','s_branch')) else None
    if m and m.group(1) in labels: succ[k].append(labels[m.group(1)])
    if op!='s_branch' and k+1<n: succ[k].append(k+1)
livein=[set() for _ in range(n)]
changed=True
it=0
while changed and it<50:
    changed=False; it+=1
    for k in range(n-1,-1,-1):
        out=set()
        for s_ in succ[k]: out|=livein[s_]
        ln,op,d,u,t=ins[k]
        new=(out-d)|u
        if new!=livein[k]: livein[k]=new; changed=True
press=[len(x) for x in livein]
mx=max(press)
print('max live',mx,'iterations',it)
for k in range(n):
    ln,op,d,u,t=ins[k]
    if op=='s_barrier' or press[k]>=max(mx-4,thr if thr else mx-4) or (thr and press[k]>=thr and k%8==0):
        print(f'{ln:6d} {press[k]:4d}  {t[:100]}')
print('--- barriers')
def rng(s):
    s=sorted(s); out=[]; a=b=None
    for x in s:
        if a is None: a=b=x
        elif x==b+1: b=x
        else: out.append((a,b)); a=b=x
    if a is not None: out.append((a,b))
    return ' '.join(f'{a}-{b}' if a!=b else f'{a}' for a,b in out)
for k in range(n):
    ln,op,d,u,t=ins[k]
    if op=='s_barrier': print(ln,press[k],rng(livein[k]))
