function mpc = case5_test
% A five-bus test network written for this repository's MATPOWER reader (tests/test_matpower.py).
% Not a PGLIB case: small made-up numbers that exercise a transformer (tap + shift), line charging, a branch without
% a thermal rating, an inactive branch, an inactive generator, a linear and a quadratic cost, shunts and two loads.
mpc.version = '2';
mpc.baseMVA = 100.0;

%% bus data
%	bus_i	type	Pd	Qd	Gs	Bs	area	Vm	Va	baseKV	zone	Vmax	Vmin
mpc.bus = [
	1	3	0.0	0.0	0.0	0.0	1	1.00	0.0	230.0	1	1.10	0.90;
	2	2	120.0	40.0	0.0	5.0	1	1.00	0.0	230.0	1	1.10	0.90;
	3	1	80.0	25.0	2.0	0.0	1	1.00	0.0	230.0	1	1.05	0.95;
	10	1	0.0	0.0	0.0	0.0	1	1.00	0.0	230.0	1	1.10	0.90;
	7	2	60.0	10.0	0.0	0.0	1	1.00	0.0	230.0	1	1.10	0.90;
];

%% generator data
%	bus	Pg	Qg	Qmax	Qmin	Vg	mBase	status	Pmax	Pmin
mpc.gen = [
	1	0.0	0.0	150.0	-150.0	1.0	100.0	1	200.0	10.0;
	2	0.0	0.0	80.0	-60.0	1.0	100.0	1	120.0	0.0;
	3	0.0	0.0	50.0	-50.0	1.0	100.0	0	90.0	0.0;
	7	0.0	0.0	100.0	-100.0	1.0	100.0	1	150.0	20.0;
];

%% generator cost data
%	2	startup	shutdown	n	c(n-1)	...	c0
mpc.gencost = [
	2	0.0	0.0	3	0.02	14.0	100.0;
	2	0.0	0.0	2	20.0	50.0	0.0;
	2	0.0	0.0	3	0.01	10.0	0.0;
	2	0.0	0.0	3	0.00	30.0	10.0;
];

%% branch data
%	fbus	tbus	r	x	b	rateA	rateB	rateC	ratio	angle	status	angmin	angmax
mpc.branch = [
	1	2	0.010	0.100	0.020	250.0	250.0	250.0	0.0	0.0	1	-30.0	30.0;
	1	3	0.020	0.150	0.030	150.0	150.0	150.0	0.0	0.0	1	-30.0	30.0;
	2	10	0.005	0.050	0.000	0.0	0.0	0.0	1.05	3.0	1	-20.0	20.0;
	10	7	0.015	0.120	0.025	180.0	180.0	180.0	0.0	0.0	1	-30.0	30.0;
	3	7	0.030	0.200	0.040	100.0	100.0	100.0	0.0	0.0	0	-30.0	30.0;
	3	10	0.012	0.090	0.015	160.0	160.0	160.0	0.0	0.0	1	-30.0	30.0;
];
