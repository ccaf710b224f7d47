import re,sys
def func(path,name):
    out=[];on=False
    for l in open(path):
        if re.match(r'^_Z\w*%s\w*:'%name,l): on=True; continue
        if on and l.startswith('.Lfunc_end'): break
        if on:
            l=re.sub(r';.*$','',l).strip()
            if l: out.append(l)
    return out
def regs(tok):
    m=re.match(r'v\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
def analyze(path,fn):
    f=func(path,fn)
    res=[]
    for i,l in enumerate(f):
        if '_dpp' not in l: continue
        ops=[o.strip() for o in l.split(None,1)[1].split(',')]
        dst=regs(ops[0]); 
        srcs=set()
        for o in ops[1:]:
            srcs|=regs(o.split()[0])
        # for v_mov_b32_dpp: old = dst
        need=srcs|dst
        # walk back
        ws=0; found=None
        for j in range(i-1,max(i-12,-1),-1):
            p=f[j]
            if p.endswith(':'): found=('label',p); break
            mn=p.split()[0]
            if mn=='s_nop':
                ws+=int(p.split()[1])+1; continue
            if mn.startswith('v_') and not mn.startswith('v_cmp'):
                pops=[o.strip() for o in p.split(None,1)[1].split(',')]
                w=regs(pops[0].split()[0])
                if w & need:
                    found=(ws,p,'SRC' if w&srcs else 'OLD'); break
            ws+=1
        res.append((i,l,found))
    return res
for path in sys.argv[1:]:
    print("=====",path)
    for i,l,found in analyze(path,'compute_starts'):
        if found and found[0]!='label' and found[0]<=2:
            print(i,l,' <= ',found)
