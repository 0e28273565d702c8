"""Per-parameter gradient error of the HIP U-Net vs an fp64 CPU run, next to torch-CPU-fp32's error (diagnostic)."""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops
from mri_epilepsy_diagnosis_amd.unet import UNet
from oracle import losses, unet_recon
from util import seeded_rand, seeded_randn

c0 = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shape = tuple(int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (32, 32, 32)
torch.manual_seed(0)
o32 = unet_recon.UNetRecon(out_channels_first_layer=c0)
o64 = copy.deepcopy(o32).double()
prod = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=c0,
            normalization="batch", upsampling_type="linear", padding=True, activation="PReLU")
prod.load_state_dict(o32.state_dict()); prod.cuda()
x = seeded_randn(5, (1, 1, *shape)); t = (seeded_rand(6, (1, 1, *shape)) < 0.1).float()
lo32 = o32(x); losses.softmax_dice_loss(lo32, t).backward()
lo64 = o64(x.double()); losses.softmax_dice_loss(lo64, t.double()).backward()
lop = prod(x.cuda()); ops.softmax_dice_loss(lop, t.cuda()).backward()
rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
print("logits: hip %.2e cpu %.2e" % (rel(lop.detach(), lo64.detach()), rel(lo32.detach(), lo64.detach())))
for (k, p64), p32, pp in zip(o64.named_parameters(), o32.parameters(), prod.parameters()):
    if ".block." in k: continue
    den = p64.grad.abs().max().item()
    print("%-55s |g| %.2e  hip %.2e  cpu %.2e" % (k, den, (pp.grad.cpu().double() - p64.grad).abs().max().item() / den,
                                                  (p32.grad.double() - p64.grad).abs().max().item() / den))
