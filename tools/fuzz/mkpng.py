import zlib, struct, os, random, sys
random.seed(int(sys.argv[1]))
out='pngs'; os.makedirs(out, exist_ok=True)
for f in os.listdir(out): os.remove(os.path.join(out,f))
def chunk(t, d, badcrc=False):
    c = zlib.crc32(t+d) & 0xffffffff
    if badcrc: c ^= 1
    return struct.pack('>I', len(d)) + t + d + struct.pack('>I', c)
def make(i):
    w = random.choice([1,2,3,7,8,9,31,64,random.randint(1,70),random.randint(0,3)]); h = random.choice([1,2,5,8,9,33,random.randint(1,40),random.randint(0,2)])
    ct,bd = random.choice([(0,1),(0,2),(0,4),(0,8),(0,16),(2,8),(2,16),(3,1),(3,2),(3,4),(3,8),(4,8),(4,16),(6,8),(6,16)])
    if random.random()<0.05: ct = random.randint(0,7)
    if random.random()<0.05: bd = random.randint(0,17)
    il = random.choice([0,0,1,1, random.randint(0,3)])
    ch = {0:1,2:3,3:1,4:2,6:4}.get(ct,1)
    bpp_bits = ch*bd
    def rawsize(w,h):
        rb = (w*bpp_bits+7)//8
        return (rb+1)*h
    if il==1:
        tot=0
        for (x0,y0,dx,dy) in [(0,0,8,8),(4,0,8,8),(0,4,4,8),(2,0,4,4),(0,2,2,4),(1,0,2,2),(0,1,1,2)]:
            pw=(w-x0+dx-1)//dx if w>x0 else 0; ph=(h-y0+dy-1)//dy if h>y0 else 0
            if pw>0 and ph>0: tot+=rawsize(pw,ph)
        n=tot
    else: n=rawsize(w,h)
    n = max(0, n + random.choice([0]*12+[-1,1,-7,13, -n//2 if n else 0]))
    raw = bytearray(random.getrandbits(8) for _ in range(n))
    # filter bytes mostly valid
    rb = (w*bpp_bits+7)//8 + 1
    if rb>1:
        for k in range(0,len(raw),rb): raw[k] = random.choice([0,1,2,3,4,4,3, random.randint(0,255) if random.random()<0.05 else 2])
    z = zlib.compress(bytes(raw), random.choice([0,1,6]))
    if random.random()<0.04: z = z[:random.randint(0,len(z))]
    if random.random()<0.03 and len(z): z = bytearray(z); z[random.randrange(len(z))] ^= 0xff; z=bytes(z)
    body = b''
    ihdr = struct.pack('>IIBBBBB', w if random.random()>0.03 else random.choice([0x7fffffff,0x80000000,0xffffffff,100000]), h if random.random()>0.03 else random.choice([0x7fffffff,0xffffffff,100000]), bd, ct, random.choice([0]*30+[1]), random.choice([0]*30+[1]), il)
    if random.random()<0.02: ihdr = ihdr[:random.randint(0,12)]
    body += chunk(b'IHDR', ihdr)
    if ct==3 or random.random()<0.1:
        npal = random.choice([0,1,2,16,256,257,random.randint(0,300)])
        pal = bytes(random.getrandbits(8) for _ in range(npal*3 + random.choice([0,0,0,1,2])))
        if random.random()<0.8: body += chunk(b'PLTE', pal)
    if random.random()<0.3:
        body += chunk(b'tRNS', bytes(random.getrandbits(8) for _ in range(random.choice([0,1,2,6,7,256,300]))))
    if random.random()<0.1: body += chunk(b'gAMA', b'\0\0\0\1')
    # split IDAT
    parts = random.choice([1,1,2,5])
    if parts==1: body += chunk(b'IDAT', z, badcrc=random.random()<0.02)
    else:
        cuts = sorted(random.randint(0,len(z)) for _ in range(parts-1))
        prev=0
        for c in cuts+[len(z)]:
            body += chunk(b'IDAT', z[prev:c]); prev=c
    if random.random()<0.9: body += chunk(b'IEND', b'')
    data = b'\x89PNG\r\n\x1a\n' + body
    if random.random()<0.05: data = data[:random.randint(0,len(data))]
    open(os.path.join(out,'%05d.png'%i),'wb').write(data)
for i in range(int(sys.argv[2])): make(i)
