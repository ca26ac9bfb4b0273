"""python -m cfl.bin.eval -- alias of cfl.bin.evaluate_total (the reference has no
cfl.bin.eval module; its evaluation is predict[_dist] + evaluate_total)."""
from .evaluate_total import main

if __name__ == '__main__':
    main()
