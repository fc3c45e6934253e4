from _dbg_common import *
B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
text = zoo.cfg_text('mobilenetv2-fpn')
net = DetectionModel(io.StringIO(text)); seeded_fill(net, 3)
ref = OracleNet(io.StringIO(text)); ref.load_state_dict(net.state_dict())
net = net.cuda(); net.train(); ref.train()
x = seeded_images(B, H, W, 5); tgt = targets(B, H, W, 20, 7)
out = net(x.cuda(), tuple(t.cuda() for t in tgt)); out['loss'].mean().backward()
want = ref(x, tgt); want['loss'].mean().backward()
print('loss', out['loss'].item(), want['loss'].item())
for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref.named_parameters()):
    a, b = p1.grad.cpu().double(), p2.grad.double()
    e = ((a - b).norm() / (b.norm() + 1e-30)).item()
    print('%-40s %-22s relerr %.3e  |g| %.3e' % (n1, tuple(p1.shape), e, b.norm().item()))
