import sys,re
lines=[l.split(';')[0].strip() for l in open(sys.argv[1])]
lo=int(sys.argv[2]); hi=int(sys.argv[3])
out=[];
def cls(op):
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_read'): return 'r'
    if op.startswith('ds_write'): return 'w'
    if op.startswith('global_load') or op.startswith('buffer_load'): return 'G'
    if op.startswith('global_store'): return 'S'
    if op.startswith('v_exp') or op.startswith('v_rcp'): return 't'
    if op.startswith('v_'): return 'v'
    if op.startswith('s_barrier'): return '|\n'
    if op.startswith('s_waitcnt'): return '_'
    if op.startswith('s_'): return 's'
    return ''
s=''
for l in lines[lo-1:hi]:
    if not l or l.endswith(':') or l.startswith('.'): continue
    s+=cls(l.split()[0])
print(s)
