"""python -m nerf_amd --config configs/lego.txt   (the upstream `python run_nerf.py --config ...`)."""
from .run_nerf import train

if __name__ == '__main__':
    train()
