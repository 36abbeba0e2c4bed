"""How often is glibc pow(h, 1/3) NOT the correctly rounded value?  (STRICT and pow; runs on any CPU, ~40 s)"""
import math, random, struct
from decimal import Decimal, getcontext
getcontext().prec = 60
Y = Decimal(1.0/3.0)          # the double nearest 1/3, exactly
def ulp_bits(a): return struct.unpack('<q', struct.pack('<d', a))[0]
def cr_pow(x):
    v = (Y * Decimal(x).ln()).exp()             # 60-digit value of x^Y
    f = float(v)                                  # Decimal -> float is correctly rounded
    return f
random.seed(1)
mis = 0; n = 0; worst = 0
for i in range(200000):
    e = random.uniform(-10, 2)
    x = 10.0 ** e * random.uniform(1, 10)
    x = float(x)
    a = math.pow(x, 1.0/3.0); b = cr_pow(x)
    n += 1
    if a != b:
        mis += 1
        worst = max(worst, abs(ulp_bits(a) - ulp_bits(b)))
print("samples", n, "glibc pow != correctly rounded:", mis, "max ulp diff", worst)
