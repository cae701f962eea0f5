import sys; sys.argv=['x']
exec(open('scratch/sizes.py').read().split('for E in (4096')[0])
run(65536,4,6,0.0); run(65536,4,6,1e-5); run(65536,8,8,1e-5); run(4096,16,10,1e-5); run(32768,16,10,1e-5); run(65536,2,4,1e-5)
