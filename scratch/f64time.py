import sys; sys.argv=['x']
exec(open('scratch/sizes.py').read().split('for E in (4096')[0])
run(65536,4,6,1e-5,precision='f32_f64acc'); run(65536,4,6,1e-5,precision='f64'); run(65536,4,6,0.0,precision='f64'); run(65536,4,6,1e-5)
