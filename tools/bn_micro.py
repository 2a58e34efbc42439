import torch, time
dev='cuda'
def run(shape, native, iters=20):
    x=torch.randn(*shape,device=dev,dtype=torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_(True)
    bn=torch.nn.BatchNorm2d(shape[1]).to(dev).to(memory_format=torch.channels_last)
    g=torch.randn_like(x)
    def step():
        with torch.autocast('cuda',dtype=torch.bfloat16):
            if native:
                with torch.backends.cudnn.flags(enabled=False):
                    y=torch.relu(bn(x))
            else:
                y=torch.relu(bn(x))
        y.backward(g)
    for _ in range(3): step()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/iters*1e3
for shape in [(24,64,128,352),(24,256,64,176),(24,512,32,88),(4,128,180,180),(4,256,90,90)]:
    nbytes=2*torch.tensor(shape).prod().item()
    print(shape, 'MB',nbytes/1e6, 'miopen ms %.3f'%run(shape,False), 'native ms %.3f'%run(shape,True))
